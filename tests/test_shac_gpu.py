"""Critic side of SHAC through the C ABI: TD(lambda) targets bit-exact against the vectors recorded from the reference, critic
values / regression gradient against the oracle's autograd (fp32; tolerances written at each comparison), and the host
mirror's `update_critic` + `soft_update` against the run recorded from the reference's own methods."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import synth  # noqa: E402
from oracle import shac_oracle as S  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'g13_shac_critic.npz'))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def targets_hip(rew, done, nv, gamma, lam, method):
    from mineral_amd.agents.diffrl import compute_target_values
    return compute_target_values(rew.to(DEV), done.to(DEV), nv.to(DEV), gamma, lam, method).cpu()


@pytest.mark.parametrize('case', sorted(synth.SHAC_CASES))
def test_target_values_bit_exact_vs_reference(case):
    obs, rew, done, nv = synth.shac_inputs(case)
    for method in ('td-lambda', 'one-step'):
        got = targets_hip(rew, done, nv, 0.99, 0.95, method)
        assert np.array_equal(got.numpy(), G[f'{case}_targets_{method}']), method


@pytest.mark.parametrize('N,T,gamma,lam', [(1000, 32, 0.99, 0.95), (4096, 16, 0.97, 0.8), (257, 1, 0.99, 0.0)])
def test_target_values_bit_exact_vs_oracle(N, T, gamma, lam):
    rew = synth.gaussianish((T, N), 5)
    nv = synth.gaussianish((T, N), 6) * 3.0
    done = (synth.lcg_t((T, N), 7, 1.0) + 0.5 < 0.1).float()
    for method in ('td-lambda', 'one-step'):
        want = S.target_values(rew, done, nv, gamma, lam, method)
        got = targets_hip(rew, done, nv, gamma, lam, method)
        assert torch.equal(got, want), method


@pytest.mark.parametrize('case', sorted(synth.SHAC_CASES))
def test_entropy_target_values_bit_exact_vs_reference(case):
    from mineral_amd.agents.diffrl import compute_target_values_with_entropy
    GE = np.load(os.path.join(ROOT, 'tests', 'golden', 'g14_shac_entropy_targets.npz'))
    N, T = synth.SHAC_CASES[case][:2]
    obs, rew, done, nv = synth.shac_inputs(case)
    ent = (-1.0 * (synth.gaussianish((T, N), 23) * 1.5 - 2.0)).to(DEV)
    for method in ('td-lambda', 'one-step'):
        for offset in (False, True):
            for scale in (False, True):
                got = compute_target_values_with_entropy(rew.to(DEV), done.to(DEV), nv.to(DEV), ent, 0.05, 0.99, 0.95, method, offset, scale, -3.0)
                assert np.array_equal(got.cpu().numpy(), GE[f'{case}_{method}_o{int(offset)}s{int(scale)}']), (method, offset, scale)


def test_target_values_rejects_bad_arguments():
    from mineral_amd.agents.diffrl import compute_target_values
    z = torch.zeros(4, 8, device=DEV)
    with pytest.raises(ValueError):
        compute_target_values(z, z, z, 0.99, 1.0, 'td-lambda')       # divides by 1 - lambda
    with pytest.raises(NotImplementedError):
        compute_target_values(z, z, z, 0.99, 0.95, 'monte-carlo')
    with pytest.raises(ValueError):
        compute_target_values(z, z[:2], z, 0.99, 0.95)


def make_critic(O, units, n_critics, seed):
    from mineral_amd.agents.diffrl import EnsembleCritic
    c = EnsembleCritic(O, None, n_critics=n_critics, mlp_kwargs=dict(units=units, norm_type='LayerNorm', act_type='SiLU'), device=DEV)
    pc = synth.make_params(S.critic_shapes(O, units, n_critics), seed)
    c.load_state_dict(pc)
    return c, pc


@pytest.mark.parametrize('rows,O,units,n_critics', [(300, 40, [256, 128], 2), (64, 8, [32, 32], 1), (1000, 60, [512, 256, 128], 2),
                                                     (8193, 40, [256, 128], 2)])   # one row into the split-bf16 regime: ragged last tiles and steps
def test_values_and_gradient_vs_oracle(rows, O, units, n_critics):
    from mineral_amd.agents.diffrl import CriticLearner
    c, pc = make_critic(O, units, n_critics, 31)
    obs = synth.gaussianish((rows, O), 3)
    tgt = synth.gaussianish((rows,), 4) * 2.0
    want_v = S.critic_values(pc, obs, units, n_critics)
    got_v = c(obs.to(DEV), return_type='all').cpu()
    assert got_v.shape == want_v.shape
    assert torch.allclose(got_v, want_v, rtol=1e-4, atol=2e-5)
    vmin, vavg = c(obs.to(DEV), return_type='min_and_avg')
    assert torch.allclose(vmin.cpu(), want_v.min(0).values, rtol=1e-4, atol=2e-5) and torch.allclose(vavg.cpu(), want_v.mean(0), rtol=1e-4, atol=2e-5)

    leaves = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    loss = S.critic_loss(leaves, obs, tgt, units, n_critics)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    learner = CriticLearner(c)
    got_loss = learner.compute_critic_loss({'obs': obs.to(DEV)}, tgt.to(DEV))
    assert abs(got_loss.item() - loss.item()) <= 1e-5 * abs(loss.item()) + 1e-6
    got = c.views_of(learner.grads)
    for k, g in grads.items():
        # fp32 GEMMs with a different summation order: 3e-4 of the tensor's largest gradient
        assert (got[k].cpu() - g).abs().max().item() <= 3e-4 * g.abs().max().item() + 1e-7, k


def test_gradient_non_finite_entries_become_zero():
    """update_critic replaces NaN / inf gradient entries by 0 before the optimizer step (shac.py:840-842)."""
    from mineral_amd.agents.diffrl import CriticLearner
    c, _ = make_critic(12, [64, 64], 2, 33)
    obs = synth.gaussianish((128, 12), 3).to(DEV)
    tgt = synth.gaussianish((128,), 4)
    tgt[5] = float('inf')
    learner = CriticLearner(c)
    learner.compute_critic_loss(obs, tgt.to(DEV))
    assert torch.isfinite(learner.grads).all()


def test_lerp_params_bit_exact():
    from mineral_amd import _hip
    cur = synth.gaussianish((100003,), 8)
    tar = synth.gaussianish((100003,), 9)
    want = tar.clone()
    want.mul_(0.2)
    want.add_((1.0 - 0.2) * cur)
    t = tar.to(DEV)
    _hip.check(_hip.lib().mrl_lerp_params(_hip.ptr(t), _hip.ptr(cur.to(DEV)), t.numel(), 0.2, 1.0 - 0.2, stream()), 'mrl_lerp_params')
    assert torch.equal(t.cpu(), want)


@pytest.mark.parametrize('case', sorted(synth.SHAC_CASES))
def test_update_critic_vs_reference_run(case):
    from mineral_amd.agents.diffrl import CriticDataset, CriticLearner, soft_update
    N, T, O, units, n_critics, n_batches, iters, optim_type, lr, max_norm = synth.SHAC_CASES[case]
    obs, rew, done, nv = synth.shac_inputs(case)
    targets = targets_hip(rew, done, nv, 0.99, 0.95, 'td-lambda').to(DEV)
    bs = N * T // n_batches
    ds = CriticDataset(bs, {'obs': obs.to(DEV)}, targets, drop_last=False)
    assert [len(ds), bs] == [int(x) for x in G[f'{case}_dataset_len']]
    assert np.array_equal(ds[len(ds) - 1][1].cpu().numpy(), G[f'{case}_last_batch_targets'])
    c, _ = make_critic(O, units, n_critics, 61)
    learner = CriticLearner(c, optim_type, dict(lr=lr, betas=(0.7, 0.95)), critic_iterations=iters, max_grad_norm=max_norm)
    res = learner.update_critic(ds)
    for k, v in res.items():
        # losses and norms of fp32 GEMM chains in another summation order
        assert np.allclose(torch.stack(v).cpu().numpy(), G[f'{case}_{k.replace("/", "_")}'], rtol=1e-3, atol=1e-5), k
    for k, v in c.state_dict().items():
        # parameters after iters * batches Adam steps: Adam's normalisation amplifies gradient rounding on small entries
        assert torch.allclose(v.cpu(), torch.from_numpy(G[f'{case}_critic.{k}']), rtol=2e-3, atol=2e-4), k
    tgt, _ = make_critic(O, units, n_critics, 62)
    soft_update(c, tgt, 0.2)
    for k, v in tgt.state_dict().items():
        assert torch.allclose(v.cpu(), torch.from_numpy(G[f'{case}_target.{k}']), rtol=2e-3, atol=2e-4), k


def make_critic_v1(O, units, seed):
    from mineral_amd.agents.diffrl import Critic
    c = Critic(O, None, mlp_kwargs=dict(units=units, norm_type='LayerNorm', act_type='ELU'), device=DEV)
    pc = synth.make_params(S.critic_shapes(O, units, 1, single=True), seed)
    c.load_state_dict(pc)
    return c, pc


@pytest.mark.parametrize('rows,O,units', [(300, 40, [256, 128]), (64, 8, [64, 64]), (1000, 60, [512, 256, 128]), (9000, 12, [128, 32]),
                                          (9000, 12, [256, 128]), (8200, 128, [128, 64])])   # the last two: split-bf16 weight gradients
def test_layernorm_elu_critic_values_and_gradient_vs_oracle(rows, O, units):
    """The single LayerNorm + ELU `Critic` (models.py:149-178): values, return types and the MSE gradient."""
    from mineral_amd.agents.diffrl import CriticLearner
    c, pc = make_critic_v1(O, units, 35)
    assert list(c.state_dict()) == [k for k, _ in S.critic_shapes(O, units, 1, single=True)]
    obs = synth.gaussianish((rows, O), 3)
    tgt = synth.gaussianish((rows,), 4) * 2.0
    want_v = S.critic_values(pc, obs, units, 1, 'ELU', single=True)[0]
    got_v = c(obs.to(DEV))
    assert got_v.shape == want_v.shape
    assert torch.allclose(got_v.cpu(), want_v, rtol=1e-4, atol=2e-5)
    allv = c(obs.to(DEV), return_type='all')
    assert isinstance(allv, list) and len(allv) == 1 and torch.equal(allv[0], got_v)
    vmin, vavg = c(obs.to(DEV), return_type='min_and_avg')
    assert torch.equal(vmin, got_v) and torch.equal(vavg, got_v) and vmin.data_ptr() != vavg.data_ptr()

    leaves = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    loss = S.critic_loss(leaves, obs, tgt, units, 1, 'ELU', single=True)
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))
    learner = CriticLearner(c)
    got_loss = learner.compute_critic_loss({'obs': obs.to(DEV)}, tgt.to(DEV))
    assert abs(got_loss.item() - loss.item()) <= 1e-5 * abs(loss.item()) + 1e-6
    got = c.views_of(learner.grads)
    for k, g in grads.items():
        # fp32 GEMMs with a different summation order: 3e-4 of the tensor's largest gradient
        assert (got[k].cpu() - g).abs().max().item() <= 3e-4 * g.abs().max().item() + 1e-7, k


@pytest.mark.parametrize('case', sorted(synth.SHAC_V1_CASES))
def test_update_critic_layernorm_elu_vs_reference_run(case):
    from mineral_amd.agents.diffrl import CriticDataset, CriticLearner
    G1 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g15_shac_v1_critic.npz'))
    N, T, O, units, _one, n_batches, iters, optim_type, lr, max_norm = synth.SHAC_V1_CASES[case]
    obs, rew, done, nv = synth.shac_inputs(case)
    targets = targets_hip(rew, done, nv, 0.99, 0.95, 'td-lambda').to(DEV)
    ds = CriticDataset(N * T // n_batches, {'obs': obs.to(DEV)}, targets, drop_last=False)
    c, _ = make_critic_v1(O, units, 63)
    assert np.allclose(c(obs.reshape(-1, O).to(DEV)).cpu().numpy(), G1[f'{case}_values0'], rtol=1e-4, atol=2e-5)
    learner = CriticLearner(c, optim_type, dict(lr=lr, betas=(0.7, 0.95)), critic_iterations=iters, max_grad_norm=max_norm)
    res = learner.update_critic(ds)
    for k, v in res.items():
        assert np.allclose(torch.stack(v).cpu().numpy(), G1[f'{case}_{k.replace("/", "_")}'], rtol=1e-3, atol=1e-5), k
    for k, v in c.state_dict().items():
        # tolerances as for the SiLU ensemble: Adam's normalisation amplifies gradient rounding on small entries
        assert torch.allclose(v.cpu(), torch.from_numpy(G1[f'{case}_critic.{k}']), rtol=2e-3, atol=2e-4), k


def test_update_critic_at_the_baseline_configuration():
    """BASELINE configuration 5, critic side: the single LayerNorm + ELU critic of cfgs/agent/DFlexAntSHAC.yaml on 8192 envs x 32
    steps, four 65536-row batches, Adam(betas 0.7 / 0.95), clip 1.0.  TD(lambda) targets bit-exact against the oracle; the loss
    and the gradient of the first 65536-row batch against the oracle's autograd on the same rows (3e-4 of each tensor's largest
    entry, like every other gradient test); one full update_critic + target blend stays finite and is bit-reproducible."""
    from mineral_amd.agents.diffrl import CriticDataset, CriticLearner
    N, T, O, units, NB, iters = 8192, 32, 64, [64, 64], 4, 16
    obs = synth.gaussianish((T, N, O), 71)
    rew, nv = synth.gaussianish((T, N), 72), synth.gaussianish((T, N), 73)
    done = (synth.lcg_t((T, N), 74) > 0.98).float()
    targets = targets_hip(rew, done, nv, 0.99, 0.95, 'td-lambda')
    assert torch.equal(targets.cpu(), S.target_values(rew, done, nv, 0.99, 0.95, 'td-lambda'))
    targets = targets.to(DEV)
    d_obs = obs.to(DEV)
    bs = N * T // NB
    assert bs == 65536

    def run():
        c, pc = make_critic_v1(O, units, 65)
        learner = CriticLearner(c, 'Adam', dict(lr=2e-3, betas=(0.7, 0.95)), critic_iterations=iters, max_grad_norm=1.0, target_critic_alpha=0.2)
        ds = CriticDataset(bs, {'obs': d_obs}, targets, drop_last=False)
        assert len(ds) == NB
        b_obs, b_tgt = ds[0]
        loss = learner.compute_critic_loss(b_obs, b_tgt)
        grads = {k: v.cpu().clone() for k, v in c.views_of(learner.grads).items()}
        res = learner.update_critic(ds)
        learner.update_target()
        torch.cuda.synchronize()
        return pc, float(loss), grads, res, torch.cat([c.flat, learner.critic_target.flat]).cpu()

    pc, loss, grads, res, flat = run()
    leaves = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    # CriticDataset flattens time-major (utils.py:42-43): batch 0 = the first 65536 rows of the [T * N] view
    ref_loss = S.critic_loss(leaves, obs.reshape(-1, O)[:bs], targets.cpu().reshape(-1)[:bs], units, 1, 'ELU', single=True)
    ref = dict(zip(leaves, torch.autograd.grad(ref_loss, list(leaves.values()))))
    assert abs(loss - ref_loss.item()) <= 1e-5 * abs(ref_loss.item()) + 1e-6
    for k, g in ref.items():
        assert (grads[k] - g).abs().max().item() <= 3e-4 * g.abs().max().item() + 1e-7, k
    assert all(torch.isfinite(torch.stack(v)).all() for v in res.values()) and torch.isfinite(flat).all()
    assert len(res['value_loss']) == iters
    assert torch.equal(flat, run()[4]), 'update_critic is deterministic'


def test_critic_rejects_unknown_activation():
    from mineral_amd import _hip
    from mineral_amd.agents.diffrl import Critic
    with pytest.raises(NotImplementedError):
        Critic(12, None, mlp_kwargs=dict(units=[64, 64], norm_type='LayerNorm', act_type='Tanh'), device=DEV)
    net = _hip.make_critic_net(12, 1, [64, 64])
    net.act_type = 7
    assert _hip.lib().mrl_critic_param_count(net) < 0


@pytest.mark.parametrize('rows,O,units', [(9000, 128, [256, 128]), (20000, 64, [512, 256, 128])])
def test_split_bf16_gemm_gradients_against_an_fp64_oracle(rows, O, units):
    """Above 8,192 rows the forward, dgrad and wgrad GEMMs of these layers multiply exact three-term bf16 splits of their fp32
    operands.  Yardstick: the same gradient computed by torch on the CPU in fp32; truth: the oracle in fp64.  The device result
    has to be at least as close to the truth as the fp32 CPU result, and within 2e-6 of each tensor's largest entry."""
    from mineral_amd.agents.diffrl import CriticLearner
    c, pc = make_critic(O, units, 2, 31)
    obs = synth.gaussianish((rows, O), 3)
    tgt = synth.gaussianish((rows,), 4) * 2.0

    def oracle_grads(dtype):
        leaves = {k: v.to(dtype).clone().requires_grad_(True) for k, v in pc.items()}
        loss = S.critic_loss(leaves, obs.to(dtype), tgt.to(dtype), units, 2)
        return dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))

    g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
    learner = CriticLearner(c)
    learner.compute_critic_loss({'obs': obs.to(DEV)}, tgt.to(DEV))
    got = c.views_of(learner.grads)
    err_dev = err_cpu = 0.0
    for k, ref in g64.items():
        scale = ref.abs().max().item()
        err_dev = max(err_dev, (got[k].cpu().double() - ref).abs().max().item() / scale)
        err_cpu = max(err_cpu, (g32[k].double() - ref).abs().max().item() / scale)
    assert err_dev <= 2e-6, (err_dev, err_cpu)
    assert err_dev <= err_cpu, (err_dev, err_cpu)


def test_misaligned_workspace_is_rejected():
    """Workspaces hold 16-byte vectors and doubles at offsets the library computes from the base: a base that is not
    256-byte aligned is refused before anything is launched."""
    from mineral_amd import _hip
    c, _ = make_critic(12, [64, 64], 1, 3)
    obs = synth.gaussianish((64, 12), 3).to(DEV)
    out = torch.empty(1, 64, dtype=torch.float32, device=DEV)
    need = int(_hip.lib().mrl_critic_workspace_bytes(c.net, 64))
    ws = torch.empty(need + 256, dtype=torch.uint8, device=DEV)
    rc = _hip.lib().mrl_critic_values(c.net, _hip.ptr(c.flat), _hip.ptr(obs), 64, None, None, 0.0, 0.0, _hip.ptr(out), C.c_void_p(ws.data_ptr() + 4), need, stream())
    assert rc == -1 and b'aligned' in _hip.lib().mrl_last_error()
    _hip.check(_hip.lib().mrl_critic_values(c.net, _hip.ptr(c.flat), _hip.ptr(obs), 64, None, None, 0.0, 0.0, _hip.ptr(out), _hip.ptr(ws), need, stream()),
               'mrl_critic_values')
