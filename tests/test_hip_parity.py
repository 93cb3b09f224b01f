"""GPU parity: the HIP path (through the C ABI / the host classes that call it) against the CPU oracle and the
golden vectors of the reference.  Integer / index results must be bit-exact; fp32 results use the tolerance
written at each comparison (the GEMMs sum in a different order than torch's CPU kernels).
"""

import os
import numpy as np
import pytest
import torch

import synth
from oracle import ppo_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


@pytest.fixture(scope='module')
def hip():
    _need_gpu()
    from mineral_amd import _hip
    _hip.lib()
    return _hip


def dev(t):
    return t.to(DEV).contiguous()


def _free_port():
    """A TCP port the kernel hands out as free right now (for the rendezvous of a subprocess group)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def cpu(t):
    torch.cuda.synchronize()
    return t.detach().cpu()


def assert_close(a, b, rtol, atol, what=''):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


NETS = {
    'mid': dict(obs=64, act=12, units=[128, 64, 32]),
    'full': dict(obs=64, act=12, units=[512, 256, 128]),
    'wide2': dict(obs=20, act=3, units=[256, 32]),
    # plain towers (no normalisation layer, ELU) as in cfgs/agent/CartpolePPO.yaml, with the state-independent log-std
    'cartpole': dict(obs=4, act=1, units=[32, 32], norm=None, actf='ELU', fixed=True),
    'elu_wide': dict(obs=20, act=3, units=[256, 64], norm=None, actf='ELU', fixed=True),
    'elu_head': dict(obs=64, act=12, units=[512, 128], norm=None, actf='ELU', fixed=False),
    'ln_fixed': dict(obs=64, act=12, units=[128, 64], fixed=True),
    # the state-independent log-std on batches above 8192 rows: in the one-launch tail (128-wide last layer) and in the heads epilogue
    'full_fixed': dict(obs=64, act=12, units=[512, 256, 128], fixed=True),
    'ln_fixed_wide': dict(obs=64, act=12, units=[256, 64], fixed=True),
    'obs128': dict(obs=128, act=12, units=[128, 128]),
    'odd_heads': dict(obs=64, act=3, units=[256, 128]),     # 6-row heads: the tensors behind them start on odd offsets
    # LayerNorm + ELU towers (mlp_kwargs of DFlexAntSHAC.yaml / DFlexAntBPTT.yaml) in the PPO ActorCritic
    'mid_lnelu': dict(obs=64, act=12, units=[128, 64, 32], actf='ELU'),
    'full_lnelu': dict(obs=64, act=12, units=[512, 256, 128], actf='ELU'),
    # the other norm_type / act_type combinations of mineral/nets/mlp.py:5-22 (chain kernels at every batch size)
    'mid_silu': dict(obs=64, act=12, units=[128, 64, 32], norm=None, actf='SiLU'),
    'full_silu': dict(obs=64, act=12, units=[512, 256, 128], norm=None, actf='SiLU'),
    'mid_lnrelu': dict(obs=64, act=12, units=[128, 64, 32], actf='ReLU'),
    'full_relu': dict(obs=64, act=12, units=[512, 256, 128], norm=None, actf='ReLU', fixed=True),
    'full_lntanh': dict(obs=64, act=12, units=[512, 256, 128], actf='Tanh'),
    'mid_tanh': dict(obs=20, act=3, units=[256, 64], norm=None, actf='Tanh'),
    # the observation width of DFlex Ant (37, padded to 40 by the host classes) in front of the DFlexAntPPO layers
    'full_obs40': dict(obs=40, act=8, units=[512, 256, 128]),
}


def oracle_cfg(spec, dist='dreamerv3_normal'):
    return O.NetConfig(spec['obs'], spec['act'], spec['units'], spec.get('norm', 'LayerNorm'), spec.get('actf', 'SiLU'), spec.get('fixed', False),
                       dist, 0.1, 1.0)


def oracle_params(spec, seed):
    shapes = synth.actor_critic_shapes(spec['obs'], spec['act'], spec['units'], spec.get('norm', 'LayerNorm') is not None, True,
                                       spec.get('fixed', False))
    p = synth.make_params(shapes, seed)
    if 'sigma' in p:
        p['sigma'] = p['sigma'] * 0.3 - 0.4      # a log-std away from its zero initial value
    return p


def make_model(spec, params, dist='dreamerv3_normal'):
    from mineral_amd.agents.ppo.models import ActorCritic
    m = ActorCritic({'obs': (spec['obs'],)}, spec['act'],
                    mlp_kwargs=dict(units=spec['units'], norm_type=spec.get('norm', 'LayerNorm'), act_type=spec.get('actf', 'SiLU')),
                    fixed_sigma=spec.get('fixed', False), actor_dist_kwargs=dict(dist_type=dist, minstd=0.1, maxstd=1.0), device=DEV)
    m.load_state_dict(params)
    return m


def make_buffer(spec, N, T, mb=None):
    from mineral_amd.agents.ppo.experience import ExperienceBuffer
    return ExperienceBuffer(N, T, N * T, mb or N * T, {'obs': (spec['obs'],)}, spec['act'], DEV)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('T,N', [(4, 8), (32, 64), (32, 1000), (7, 333), (1, 5)])
def test_gae_flatten_advnorm_valuenorm(hip, golden, T, N):
    from mineral_amd.common.normalizers import RunningMeanStd
    spec = NETS['mid']
    buf = make_buffer(spec, N, T)
    rewards, values = synth.gaussianish((T, N, 1), 11), synth.gaussianish((T, N, 1), 12)
    dones = (synth.lcg_t((T, N), 14) > 0.4).to(torch.uint8)
    neglogp = synth.gaussianish((T, N), 13)
    last_values = synth.gaussianish((N, 1), 18)
    s = buf.storage_dict
    s['rewards'].copy_(rewards); s['values'].copy_(values); s['dones'].copy_(dones); s['neglogp'].copy_(neglogp)
    vr = RunningMeanStd((1,), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    buf.compute_return(dev(last_values), 0.99, 0.95, vr)
    d = buf.prepare_training()
    # oracle
    ret, _ = O.gae_returns(rewards, values, dones, last_values, 0.99, 0.95)
    fr, fv = O.flatten_env_major(ret), O.flatten_env_major(values)
    adv = O.normalize_advantages(fr, fv) if N * T > 1 else None
    st = O.rms_init((1,))
    st = O.rms_update(st, fv); nv = O.rms_normalize(st, fv)
    st = O.rms_update(st, fr); nr = O.rms_normalize(st, fr)
    # the un-normalised GAE returns are bit-exact: check through the inverse of the value normalisation is not
    # possible, so run once more without value_rms
    buf2 = make_buffer(spec, N, T)
    for k in ('rewards', 'values', 'dones', 'neglogp'):
        buf2.storage_dict[k].copy_(s[k])
    buf2.compute_return(dev(last_values), 0.99, 0.95, None)
    d2 = buf2.prepare_training()
    assert torch.equal(cpu(d2['returns']), fr), 'GAE returns must be bit-exact'
    assert torch.equal(cpu(d2['values']), fv)
    assert torch.equal(cpu(d2['neglogp']), O.flatten_env_major(neglogp)), 'flat index map n*T+t'
    assert torch.equal(cpu(d2['dones']), O.flatten_env_major(dones))
    if adv is not None and N * T > 2:
        assert_close(cpu(d2['advantages']), adv, 2e-6, 2e-6, 'advantages')
        assert_close(cpu(d['advantages']), adv, 2e-6, 2e-6, 'advantages')
        assert_close(cpu(d['values']), nv, 2e-6, 2e-6, 'normalised values')
        assert_close(cpu(d['returns']), nr, 2e-6, 2e-6, 'normalised returns')
        assert_close(cpu(vr.running_mean), st['running_mean'], 1e-6, 1e-9)
        assert_close(cpu(vr.running_var), st['running_var'], 1e-6, 1e-9)
        assert cpu(vr.running_count).item() == st['running_count'].item()
    if (T, N) == (32, 64):
        g = golden('g1_gae_mid')
        assert np.array_equal(cpu(d2['returns']).numpy(), g['flat_returns'])
        assert_close(cpu(d2['advantages']), g['advantages'], 2e-6, 2e-6)


def test_running_mean_std_golden(hip, golden):
    from mineral_amd.common.normalizers import RunningMeanStd
    g = golden('g3_rms_ppo')
    rms = RunningMeanStd((6,), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    for i in range(5):
        x = synth.gaussianish((37 + 11 * i, 6), 100 + i) * (1.0 + i) + 0.3 * i
        rms.update(dev(x))
        assert_close(cpu(rms.running_mean), g[f'mean{i}'], 1e-6, 1e-8, f'mean{i}')
        assert_close(cpu(rms.running_var), g[f'var{i}'], 2e-6, 1e-8, f'var{i}')
        assert cpu(rms.running_count).item() == float(g[f'count{i}'])
        assert_close(cpu(rms.normalize(dev(x * 3))), g[f'norm{i}'], 2e-6, 2e-6, f'norm{i}')
        assert_close(cpu(rms.unnormalize(dev(x * 3))), g[f'unnorm{i}'], 2e-6, 2e-6, f'unnorm{i}')


def test_batch_moments_groups(hip):
    x = synth.gaussianish((4096, 64), 5) * 3 + 1.5
    L = hip.lib()
    mean = torch.empty(8, 64, device=DEV); var = torch.empty(8, 64, device=DEV)
    ws = torch.empty(8 * 1024 * 2 * 64 * 8, dtype=torch.uint8, device=DEV)
    d_x = dev(x)
    hip.check(L.mrl_batch_moments(hip.ptr(d_x), 4096, 64, 512, 1, hip.ptr(mean), hip.ptr(var), hip.ptr(ws), ws.numel(), None), 'm')
    xr = x.view(8, 512, 64)
    assert_close(cpu(mean), xr.mean(1), 1e-6, 1e-7)
    assert_close(cpu(var), xr.var(1, correction=1), 2e-6, 1e-7)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('net,N,dist', [('mid', 64, 'dreamerv3_normal'), ('full', 16, 'dreamerv3_normal'), ('full', 300, 'normal'),
                                       ('wide2', 129, 'dreamerv3_normal'), ('cartpole', 100, 'normal'), ('elu_wide', 300, 'normal'),
                                       ('elu_head', 77, 'dreamerv3_normal'), ('ln_fixed', 64, 'normal'),
                                       ('mid_lnelu', 16, 'dreamerv3_normal'), ('full_lnelu', 300, 'normal'),
                                       ('mid_silu', 70, 'dreamerv3_normal'), ('full_relu', 300, 'normal'), ('full_lntanh', 9000, 'dreamerv3_normal'),
                                       ('mid_tanh', 33, 'normal'),
                                       # above 8192 rows: the stationary-weight first-layer forward, whole and ragged row counts, 64 and 40 inputs
                                       ('full', 8448, 'dreamerv3_normal'), ('full_obs40', 8333, 'normal')])
def test_policy_act(hip, golden, net, N, dist):
    from mineral_amd.common.normalizers import RunningMeanStd
    spec = NETS[net]
    T = 3
    params = oracle_params(spec, 3)
    cfg = oracle_cfg(spec, dist)
    model = make_model(spec, params, dist)
    buf = make_buffer(spec, N, T)
    obs = synth.gaussianish((N, spec['obs']), 41)
    noise = synth.gaussianish((N, spec['act']), 43)
    obs_rms = RunningMeanStd((spec['obs'],), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    value_rms = RunningMeanStd((1,), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    xs = synth.gaussianish((50, spec['obs']), 77) * 1.5 + 0.2
    vs = synth.gaussianish((50, 1), 78) * 2.0 + 0.5
    obs_rms.update(dev(xs)); value_rms.update(dev(vs))
    o_obs = O.rms_update(O.rms_init((spec['obs'],)), xs)
    o_val = O.rms_update(O.rms_init((1,)), vs)
    L = hip.lib()
    ws = torch.empty(int(L.mrl_workspace_bytes(model.net, N)), dtype=torch.uint8, device=DEV)
    clamped = torch.empty(N, spec['act'], device=DEV)
    t = 1
    d_obs, d_noise = dev(obs), dev(noise)     # keep the device copies alive while kernels use them
    hip.check(L.mrl_policy_act(model.net, hip.ptr(model.flat), obs_rms.c_struct(), value_rms.c_struct(), hip.ptr(d_obs),
                               hip.ptr(d_noise), t, buf.c, hip.ptr(clamped), None, hip.ptr(ws), ws.numel(), None), 'act')
    ref = O.model_act(params, O.rms_normalize(o_obs, obs), noise, cfg)
    ref_v = O.rms_unnormalize(o_val, ref['values'])
    s = buf.storage_dict
    tol = dict(rtol=2e-5, atol=2e-5)
    assert torch.equal(cpu(s['obses']['obs'][:, t]), obs)
    assert_close(cpu(s['mu'][:, t]), ref['mu'], **tol, what='mu')
    assert_close(cpu(s['sigma'][:, t]), ref['sigma'], **tol, what='sigma')
    assert_close(cpu(s['actions'][:, t]), ref['actions'], **tol, what='actions')
    assert_close(cpu(clamped), ref['actions'].clamp(-1, 1), **tol, what='clamped')
    assert_close(cpu(s['neglogp'][t]), ref['neglogp'], rtol=2e-5, atol=1e-4, what='neglogp')
    assert_close(cpu(s['values'][t]), ref_v, **tol, what='values')
    assert float(s['mu'][:, 0].abs().sum()) == 0.0 and float(s['mu'][:, 2].abs().sum()) == 0.0, 'only row t is written'
    # bootstrap forward: values only
    last = torch.empty(N, device=DEV)
    hip.check(L.mrl_policy_act(model.net, hip.ptr(model.flat), obs_rms.c_struct(), value_rms.c_struct(), hip.ptr(d_obs),
                               None, T, buf.c, None, hip.ptr(last), hip.ptr(ws), ws.numel(), None), 'act boot')
    assert_close(cpu(last), ref_v.squeeze(1), **tol, what='last_values')
    if (net == 'full' or net == 'mid_lnelu') and N == 16 and dist == 'dreamerv3_normal':
        # the reference's own outputs for this network (no normalisers): golden g4
        g = golden('g4_model_full_dv3' if net == 'full' else 'g4_model_mid_lnelu')
        ident_o = RunningMeanStd((spec['obs'],), eps=0.0, with_clamp=True, initial_count=1, device=DEV)
        buf4 = make_buffer(spec, 16, 1)
        hip.check(L.mrl_policy_act(model.net, hip.ptr(model.flat), None, None, hip.ptr(d_obs), hip.ptr(d_noise), 0, buf4.c,
                                   hip.ptr(clamped), None, hip.ptr(ws), ws.numel(), None), 'act g4')
        s4 = buf4.storage_dict
        assert_close(cpu(s4['mu'][:, 0]), g['act_mu'], **tol)
        assert_close(cpu(s4['sigma'][:, 0]), g['act_sigma'], **tol)
        assert_close(cpu(s4['actions'][:, 0]), g['act_actions'], **tol)
        assert_close(cpu(s4['neglogp'][0]), g['act_neglogp'], rtol=2e-5, atol=1e-4)
        assert_close(cpu(s4['values'][0]), g['act_values'], **tol)
        del ident_o


def _fill_training_buffer(spec, buf, N, T, seed=0):
    B = N * T
    A = spec['act']
    data = {
        'obs': synth.gaussianish((B, spec['obs']), 200 + seed) * 1.3 + 0.1,
        'actions': synth.gaussianish((B, A), 201 + seed) * 0.7,
        'mu': torch.tanh(synth.gaussianish((B, A), 202 + seed) * 0.5),
        'sigma': synth.lcg_t((B, A), 203 + seed) * 0.6 + 0.5,
        'neglogp': synth.gaussianish((B,), 204 + seed) * 0.5 + 11.0,
        'advantages': synth.gaussianish((B,), 205 + seed),
        'returns': synth.gaussianish((B, 1), 206 + seed),
        'values': synth.gaussianish((B, 1), 207 + seed),
    }
    s = buf.storage_dict
    s['obses']['obs'].copy_(data['obs'].view(N, T, -1))
    s['actions'].view(B, A).copy_(data['actions'])
    s['mu'].view(B, A).copy_(data['mu'])
    s['sigma'].view(B, A).copy_(data['sigma'])
    buf._flat['neglogp'].copy_(data['neglogp'])
    buf._flat['advantages'].copy_(data['advantages'])
    buf._flat['returns'].copy_(data['returns'].squeeze(1))
    buf._flat['values'].copy_(data['values'].squeeze(1))
    return data


@pytest.mark.parametrize('net,N,T,mb,dist,clipv,btype', [
    ('mid', 64, 8, 128, 'dreamerv3_normal', 0, 'bound'),
    ('mid', 50, 4, 100, 'normal', 1, 'reg'),
    ('mid', 64, 8, 128, 'dreamerv3_normal', 2, 'bound'),     # clipv 2: smooth_clamp of the ratio (ppo.py:449-457)
    ('full', 64, 8, 256, 'dreamerv3_normal', 0, 'bound'),
    ('wide2', 37, 8, 148, 'dreamerv3_normal', 0, 'bound'),
    ('full', 32, 32, 1024, 'dreamerv3_normal', 0, 'bound'),
    ('cartpole', 64, 8, 128, 'normal', 0, 'bound'),
    ('cartpole', 37, 8, 296, 'normal', 1, 'reg'),
    ('elu_wide', 50, 8, 200, 'normal', 0, 'bound'),
    ('elu_head', 64, 8, 256, 'dreamerv3_normal', 0, 'bound'),
    ('ln_fixed', 64, 8, 256, 'normal', 0, 'bound'),
    ('mid_lnelu', 64, 8, 128, 'dreamerv3_normal', 0, 'bound'),
    ('full_lnelu', 64, 8, 256, 'normal', 1, 'reg'),
    ('full_lnelu', 1041, 8, 8328, 'dreamerv3_normal', 0, 'bound'),     # split-bf16 kernels with the ELU prologue / derivative
    # minibatches of 8192 rows and more take the weight gradients of whole-128 layers through the split-bf16 kernel; 8328 rows
    # end in a partial 16-row step
    ('full', 1041, 8, 8328, 'dreamerv3_normal', 0, 'bound'),
    ('full_obs40', 1041, 8, 8328, 'dreamerv3_normal', 0, 'bound'),   # first layer on fewer than 64 inputs (one-launch LayerNorm backward + weight gradient)
    ('elu_head', 1041, 8, 8328, 'dreamerv3_normal', 0, 'bound'),
    ('full_fixed', 1041, 8, 8328, 'normal', 0, 'bound'),
    ('ln_fixed_wide', 1041, 8, 8328, 'normal', 1, 'reg'),
    ('obs128', 1024, 8, 8192, 'normal', 0, 'bound'),
    ('odd_heads', 1041, 8, 8328, 'dreamerv3_normal', 0, 'bound'),
    # no norm + SiLU, LayerNorm / no norm + ReLU, LayerNorm / no norm + Tanh: small batches and batches above 8192 rows
    ('mid_silu', 64, 8, 128, 'dreamerv3_normal', 0, 'bound'),
    ('full_silu', 1041, 8, 8328, 'dreamerv3_normal', 0, 'bound'),
    ('mid_lnrelu', 37, 8, 296, 'normal', 1, 'reg'),
    ('full_relu', 64, 8, 256, 'normal', 0, 'bound'),
    ('full_relu', 1041, 8, 8328, 'normal', 0, 'bound'),
    ('full_lntanh', 64, 8, 256, 'dreamerv3_normal', 0, 'bound'),
    ('full_lntanh', 1041, 8, 8328, 'dreamerv3_normal', 0, 'bound'),
    ('mid_tanh', 50, 8, 200, 'normal', 0, 'bound'),
    # the benchmark's own launches: the DFlexAntPPO network on 32768-row minibatches (256-block weight-gradient plans, 256-row
    # tiles, 256 row tiles per tower), as the only minibatch of the batch and as the second of two (row0 = 32768)
    ('full', 1024, 32, 32768, 'dreamerv3_normal', 0, 'bound'),
    ('full', 2048, 32, 32768, 'dreamerv3_normal', 0, 'bound'),
])
def test_ppo_grad_and_adam(hip, net, N, T, mb, dist, clipv, btype):
    spec = NETS[net]
    params = oracle_params(spec, 9)
    cfg = oracle_cfg(spec, dist)
    model = make_model(spec, params, dist)
    buf = make_buffer(spec, N, T, mb)
    data = _fill_training_buffer(spec, buf, N, T)
    smooth, clipv = clipv == 2, clipv & 1
    hp = O.PPOHParams(horizon_len=T, minibatch_size=mb, clip_value_loss=bool(clipv), bounds_type=btype, entropy_coef=0.01,
                      bounds_loss_coef=0.05 if dist == 'normal' else 1e-4, use_smooth_clamp=smooth)
    chp = hip.PpoHp(hp.e_clip, hp.critic_coef, hp.entropy_coef, hp.bounds_loss_coef, 0 if btype == 'bound' else 1, clipv, int(smooth))
    L = hip.lib()
    ws = torch.empty(int(L.mrl_workspace_bytes(model.net, mb)), dtype=torch.uint8, device=DEV)
    grads = torch.full((model.param_count,), float('nan'), device=DEV)
    metrics = torch.zeros(hip.NMETRIC, dtype=torch.float64, device=DEV)
    # normaliser vectors as the merge kernel emits them
    o_rms = O.rms_update(O.rms_init((spec['obs'],)), data['obs'][:mb])
    mean_f = dev(o_rms['running_mean'].float())
    std_f = dev(torch.sqrt(o_rms['running_var'].float() + 1e-5))
    row0 = (N * T // mb - 1) * mb     # the last minibatch
    sl = slice(row0, row0 + mb)
    hip.check(L.mrl_ppo_grad(model.net, hip.ptr(model.flat), chp, buf.c, row0, mb, hip.ptr(mean_f), hip.ptr(std_f), -5.0, 5.0,
                             hip.ptr(grads), hip.ptr(metrics), hip.ptr(ws), ws.numel(), None), 'grad')
    # oracle on the same minibatch
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    mbd = {k: v[sl] for k, v in data.items()}
    out = O.model_forward(leaves, O.rms_normalize(o_rms, mbd['obs']), mbd['actions'], cfg)
    loss, diag = O.total_loss(out, mbd, hp)
    g = torch.autograd.grad(loss, list(leaves.values()))
    g = dict(zip(leaves.keys(), g))
    kl = O.policy_kl(out['mu'].detach(), out['sigma'].detach(), mbd['mu'], mbd['sigma'])
    gv = model.views_of(grads)
    torch.cuda.synchronize()
    assert not torch.isnan(grads).any(), 'every gradient element must be written'
    for name, ref in g.items():
        got = cpu(gv[name])
        scale = float(ref.abs().max()) + 1e-12
        err = float((got - ref).abs().max())
        assert err <= 3e-4 * scale + 1e-9, f'{name}: max err {err:.3e} vs scale {scale:.3e}'
    m = cpu(metrics)
    names = dict(zip(hip.METRIC_NAMES, m.tolist()))
    assert_close(names['loss/actor'], float(diag['loss/actor']), 2e-4, 1e-6, 'a_loss')
    assert_close(names['loss/critic'], float(diag['loss/critic']), 2e-4, 1e-6, 'c_loss')
    assert_close(names['loss/entropy'], float(diag['loss/entropy']), 2e-5, 1e-6, 'entropy')
    assert_close(names['loss/bounds'], float(diag['loss/bounds']), 2e-4, 1e-7, 'b_loss')
    assert_close(names['loss/total'], float(diag['loss/total']), 2e-4, 1e-6, 'total')
    assert_close(names['clip_frac'], float(diag['clip_frac']), 0, 1.5 / mb, 'clip_frac')
    assert_close(names['explained_var'], float(diag['explained_var']), 2e-4, 1e-5, 'explained_var')
    assert_close(names['kl'], float(kl), 2e-4, 1e-6, 'kl')
    # mu / sigma write-back of exactly this minibatch's rows
    A = spec['act']
    mu_after = cpu(buf.storage_dict['mu'].view(-1, A))
    assert_close(mu_after[sl], out['mu'].detach(), 2e-5, 2e-5, 'mu write-back')
    assert torch.equal(mu_after[:row0], data['mu'][:row0]), 'rows outside the minibatch are untouched'
    assert_close(cpu(buf.storage_dict['sigma'].view(-1, A))[sl], out['sigma'].detach(), 2e-5, 2e-5, 'sigma write-back')

    # ---- clip + Adam / AdamW on the flat vectors against the oracle's per-tensor step --------------------------------
    for optim in ('AdamW', 'Adam'):
        p_ref = {k: v.clone() for k, v in params.items()}
        st = O.adam_state_init(p_ref)
        model.load_state_dict(params)
        m1 = torch.zeros_like(model.flat); m2 = torch.zeros_like(model.flat)
        gn = torch.zeros(1, device=DEV)
        scratch = torch.empty(4096, dtype=torch.uint8, device=DEV)
        gh = {k: cpu(gv[k]).clone() for k in g}
        for step in (1, 2, 3):
            clipped, total = O.clip_grad_norm(list(gh.values()), 0.5)
            O.adam_step(p_ref, dict(zip(gh.keys(), clipped)), st, 5e-4, optim, eps=1e-5)
            hip.check(L.mrl_adam_step(hip.ptr(model.flat), hip.ptr(grads), hip.ptr(m1), hip.ptr(m2), model.param_count, 5e-4, 0.9,
                                      0.999, 1e-5, 0.01 if optim == 'AdamW' else 0.0, 1 if optim == 'AdamW' else 0, step, 0.5, 1.0,
                                      hip.ptr(gn), hip.ptr(scratch), None), 'adam')
        assert_close(cpu(gn)[0], float(total), 1e-5, 1e-9, 'grad norm')
        for name, ref in p_ref.items():
            assert_close(cpu(model._views[name]), ref, 1e-5, 1e-7, f'{optim} {name}')
        mv = model.views_of(m1)
        for name in p_ref:
            assert_close(cpu(mv[name]), st['exp_avg'][name], 1e-5, 1e-10, f'{optim} exp_avg {name}')


def test_benchmark_minibatch_gradients_against_an_fp64_oracle(hip):
    """The 32768-row minibatch of the benchmark through the split-bf16 GEMMs.  Truth: the oracle evaluated in fp64; yardstick:
    the oracle in fp32 on the CPU.  The device gradient must be within 5e-6 of each tensor's largest entry of the truth
    (60 times tighter than the fp32-vs-fp32 tolerance of test_ppo_grad_and_adam; measured 1.5e-6) and at least as close
    to it as the fp32 CPU result is (measured 4.6e-6)."""
    spec = NETS['full']
    N, T, mb = 1024, 32, 32768
    params = oracle_params(spec, 9)
    cfg = oracle_cfg(spec)
    model = make_model(spec, params)
    buf = make_buffer(spec, N, T, mb)
    data = _fill_training_buffer(spec, buf, N, T)
    hp = O.PPOHParams(horizon_len=T, minibatch_size=mb, entropy_coef=0.01)
    chp = hip.PpoHp(hp.e_clip, hp.critic_coef, hp.entropy_coef, hp.bounds_loss_coef, 0, 0, 0)
    L = hip.lib()
    ws = torch.empty(int(L.mrl_workspace_bytes(model.net, mb)), dtype=torch.uint8, device=DEV)
    grads = torch.full((model.param_count,), float('nan'), device=DEV)
    metrics = torch.zeros(hip.NMETRIC, dtype=torch.float64, device=DEV)
    hip.check(L.mrl_ppo_grad(model.net, hip.ptr(model.flat), chp, buf.c, 0, mb, None, None, 0.0, 0.0, hip.ptr(grads), hip.ptr(metrics),
                             hip.ptr(ws), ws.numel(), None), 'grad')

    def oracle_grads(dtype):
        leaves = {k: v.to(dtype).clone().requires_grad_(True) for k, v in params.items()}
        mbd = {k: v.to(dtype) for k, v in data.items()}
        out = O.model_forward(leaves, mbd['obs'], mbd['actions'], cfg)
        loss, _ = O.total_loss(out, mbd, hp)
        return dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()))))

    g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
    gv = model.views_of(grads)
    err_dev = err_cpu = 0.0
    for k, ref in g64.items():
        scale = ref.abs().max().item()
        err_dev = max(err_dev, (cpu(gv[k]).double() - ref).abs().max().item() / scale)
        err_cpu = max(err_cpu, (g32[k].double() - ref).abs().max().item() / scale)
    print(f'fp64-oracle distance: device {err_dev:.3e}, fp32 CPU {err_cpu:.3e}')
    assert err_dev <= 5e-6, (err_dev, err_cpu)
    assert err_dev <= err_cpu, (err_dev, err_cpu)


# ---------------------------------------------------------------------------------------------------------------------
def _agent_cfg(spec, N, T, mb, mini_epochs, epochs, optim, logdir):
    from mineral_amd.config import Config
    ppo = dict(multi_gpu=False, num_actors=N, reward_shaper=dict(fn='scale', scale=0.5), max_agent_steps=epochs * N * T,
               horizon_len=T, minibatch_size=mb, normalize_value=True, value_bootstrap=True, clip_value_loss=False,
               normalize_advantage=True, gamma=0.99, tau=0.95, entropy_coef=0.0, e_clip=0.2, use_smooth_clamp=False,
               critic_coef=4, bounds_loss_coef=0.0001, bounds_type='bound', mini_epochs=mini_epochs, optim_type=optim,
               optim_kwargs=dict(lr=5e-4, eps=1e-5), lr_schedule='kl', kl_threshold=0.008, max_grad_norm=0.5, truncate_grads=True)
    return Config.wrap(dict(
        rl_device=DEV, sim_device=DEV, multi_gpu=False, env_render=False, run='train', seed=42,
        task=dict(name='Golden', env=dict(numEnvs=N), env_autoresets=True),
        agent=dict(algo='PPO', print_every=-1, ckpt_every=-1, tracker_len=100, metrics_kwargs={},
                   network=dict(normalize_input=True, actor_critic='ActorCritic',
                                actor_critic_kwargs=dict(separate_value_mlp=True,
                                                         mlp_kwargs=dict(units=spec['units'], norm_type=spec.get('norm', 'LayerNorm'),
                                                                         act_type=spec.get('actf', 'SiLU')),
                                                         fixed_sigma=spec.get('fixed', False),
                                                         actor_dist_kwargs=(dict(dist_type='normal') if spec.get('fixed', False) else
                                                                            dict(dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0)))),
                   ppo=ppo)))


@pytest.mark.parametrize('fixture,net,optim,mb', [('g8_train_mid_dv3_adamw', 'mid', 'AdamW', 128), ('g8_train_cartpole_adam', 'cartpole', 'Adam', 128),
                                                  ('g8_train_full_dv3_adamw', 'full', 'AdamW', 256),
                                                  ('g8_train_mid_lnelu_adamw', 'mid_lnelu', 'AdamW', 128)])
def test_end_to_end_training_vs_reference_golden(hip, golden, tmp_path, fixture, net, optim, mb):
    """PPO.train() of the product on the deterministic env against what the REFERENCE produced: the LayerNorm+SiLU net
    with a sigma head (fixture g8 mid), the network of CartpolePPO.yaml (no norm, ELU, state-independent log-std), the
    DFlexAntPPO network itself ([512, 256, 128]) and LayerNorm + ELU towers."""
    from mineral_amd.agents import PPO
    g = golden(fixture)
    spec = NETS[net]
    N, T, mini_epochs, epochs = 64, 8, 2, 2
    cfg = _agent_cfg(spec, N, T, mb, mini_epochs, epochs, optim, str(tmp_path))
    env = synth.GoldenEnv(N, spec['obs'], spec['act'], device=DEV)
    agent = PPO(cfg, logdir=str(tmp_path), accelerator=None, datasets=None, env=env)
    agent.model.load_state_dict(synth.make_params(synth.actor_critic_shapes(spec['obs'], spec['act'], spec['units'],
                                                                            spec.get('norm', 'LayerNorm') is not None, True, spec.get('fixed', False)), 5))
    agent.noise_fn = lambda k, shape: synth.gaussianish(shape, 900 + k).to(DEV)
    agent.save = lambda *a, **k: None
    agent._checkpoint_save = lambda *a, **k: None
    seen = {}
    orig = agent.train_epoch

    def spy():
        e = agent.epoch
        if e == 0:
            d = agent.storage.data_dict
            seen['pre'] = {k: cpu(d[k]).clone() for k in ('values', 'returns', 'advantages', 'neglogp', 'actions', 'mu', 'sigma',
                                                         'rewards', 'dones')}
            seen['pre']['obs'] = cpu(d['obses']['obs']).clone()
        res = orig()
        seen[f'kl{e}'] = cpu(res['avg_kl']).clone()
        seen[f'lr{e}'] = agent.last_lr
        seen[f'mu{e}'] = cpu(agent.storage.data_dict['mu']).clone()
        seen[f'obs_rms{e}'] = {k: cpu(v).clone() for k, v in agent.obs_rms['obs'].state_dict().items()}
        seen[f'val_rms{e}'] = {k: cpu(v).clone() for k, v in agent.value_rms.state_dict().items()}
        for k in ('loss/total', 'loss/actor', 'loss/critic', 'clip_frac', 'explained_var', 'grad_norm/all'):
            seen[f'{k}{e}'] = cpu(res[k]).clone()
        return res

    agent.train_epoch = spy
    agent.train()
    tol = dict(rtol=1e-4, atol=2e-5)
    pre = seen['pre']
    assert np.array_equal(pre['dones'].numpy(), g['e0_pre_dones'])
    for k in ('obs', 'actions', 'mu', 'sigma', 'rewards', 'values', 'returns', 'advantages'):
        assert_close(pre[k], g[f'e0_pre_{k}'], **tol, what=f'rollout {k}')
    assert_close(pre['neglogp'], g['e0_pre_neglogp'], rtol=1e-4, atol=2e-4, what='rollout neglogp')
    for e in range(epochs):
        assert_close(seen[f'kl{e}'], g[f'e{e}_kl'], rtol=5e-3, atol=2e-6, what=f'avg_kl epoch {e}')
        assert seen[f'lr{e}'] == float(g[f'e{e}_last_lr']), 'KL-adaptive learning rate path'
        assert_close(seen[f'mu{e}'], g[f'e{e}_data_mu'], **tol, what='mu write-back')
        assert_close(seen[f'obs_rms{e}']['running_mean'], g[f'e{e}_obs_rms_mean'], 1e-6, 2e-7)
        assert_close(seen[f'obs_rms{e}']['running_var'], g[f'e{e}_obs_rms_var'], 1e-5, 2e-7)
        assert seen[f'obs_rms{e}']['running_count'].item() == float(g[f'e{e}_obs_rms_count'])
        assert_close(seen[f'val_rms{e}']['running_mean'], g[f'e{e}_value_rms_mean'], 1e-4, 1e-6)
        for k in ('loss/total', 'loss/actor', 'loss/critic', 'explained_var', 'grad_norm/all'):
            assert_close(seen[f'{k}{e}'], g[f'e{e}_{k.replace("/", "_")}'], rtol=2e-3, atol=2e-5, what=f'{k} epoch {e}')
    last = epochs - 1
    for n, p in agent.model.named_parameters():
        assert_close(cpu(p), g[f'e{last}_p_{n}'], rtol=2e-3, atol=2e-5, what=f'final {n}')
    assert agent.agent_steps == int(g['agent_steps'])


def test_training_continues_from_a_checkpoint_written_by_the_reference(hip, golden, tmp_path):
    """tests/golden/g16_reference_checkpoint.pth is the reference's own PPO.save output after one epoch (ppo.py:416-426) and
    g16_ckpt_continued what the reference did after PPO.load of it into a fresh agent (ppo.py:429-446).  The product loads the
    same file and has to continue the same way: epoch counters, parameters, Adam moments and step, running statistics and the
    optimizer's learning rate come from the file; last_lr restarts from the configured value like the reference's."""
    from mineral_amd.agents import PPO
    g = golden('g16_ckpt_continued')
    spec = NETS['mid']
    N, T, mb, mini_epochs, epochs = 64, 8, 128, 2, 2
    cfg = _agent_cfg(spec, N, T, mb, mini_epochs, epochs, 'AdamW', str(tmp_path))
    env = synth.GoldenEnv(N, spec['obs'], spec['act'], device=DEV)
    agent = PPO(cfg, logdir=str(tmp_path), accelerator=None, datasets=None, env=env)
    agent.load(os.path.join(os.path.dirname(__file__), 'golden', 'g16_reference_checkpoint.pth'))
    assert (agent.epoch, agent.agent_steps, agent.optim_step) == (0, N * T, mini_epochs * (N * T // mb))
    agent.noise_fn = lambda k, shape: synth.gaussianish(shape, 900 + k).to(DEV)
    agent.save = lambda *a, **k: None
    agent._checkpoint_save = lambda *a, **k: None
    seen = {}
    orig = agent.train_epoch

    def spy():
        res = orig()
        seen['kl'] = cpu(res['avg_kl']).clone()
        return res

    agent.train_epoch = spy
    agent.train()
    assert agent.epoch == 1 and agent.agent_steps == int(g['agent_steps'])
    assert_close(seen['kl'], g['e1_kl'], rtol=5e-3, atol=2e-6, what='avg_kl after the resume')
    assert agent.last_lr == float(g['e1_last_lr'])
    assert cpu(agent.obs_rms['obs'].running_count).item() == float(g['e1_obs_rms_count'])
    assert_close(cpu(agent.obs_rms['obs'].running_mean), g['e1_obs_rms_mean'], 1e-6, 2e-7)
    assert_close(cpu(agent.value_rms.running_var), g['e1_value_rms_var'], 1e-4, 1e-6)
    for n, p in agent.model.named_parameters():
        assert_close(cpu(p), g[f'e1_p_{n}'], rtol=2e-3, atol=2e-5, what=f'{n} after the resumed epoch')


def test_observation_width_that_is_not_a_multiple_of_4(hip, tmp_path):
    """DFlex Ant has 37 observations and 8 actions.  The kernels read 16-byte vectors, so the host classes pad observation
    rows, the first layer's weight rows and the running statistics with zero columns; the tensors under the reference's
    names keep their true shapes.  One PPO.train() epoch of the product against the oracle's (which works on 37 columns)."""
    from mineral_amd.agents import PPO
    from test_oracle_golden import run_oracle_training
    spec = dict(obs=37, act=8, units=[128, 64])
    ospec = dict(obs=37, act=8, units=[128, 64], norm='LayerNorm', act_type='SiLU', fixed_sigma=False, dist_type='dreamerv3_normal')
    N, T, mb, mini_epochs, epochs = 64, 8, 128, 2, 1
    cfg = _agent_cfg(spec, N, T, mb, mini_epochs, epochs, 'AdamW', str(tmp_path))
    env = synth.GoldenEnv(N, 37, 8, device=DEV)
    agent = PPO(cfg, logdir=str(tmp_path), accelerator=None, datasets=None, env=env)
    shapes = synth.actor_critic_shapes(37, 8, spec['units'], True, True, False)
    agent.model.load_state_dict(synth.make_params(shapes, 5))
    sd = agent.model.state_dict()
    assert [tuple(v.shape) for v in sd.values()] == [tuple(sh) for _, sh in shapes]
    assert sd['actor_mlp.mlp.0.weight'].shape == (128, 37) and agent.model.net.obs_dim == 40
    agent.noise_fn = lambda k, shape: synth.gaussianish(shape, 900 + k).to(DEV)
    agent.save = lambda *a, **k: None
    agent._checkpoint_save = lambda *a, **k: None
    agent.train()
    want = {}

    def on_epoch(e, pre, data, res, lr, obs_rms, value_rms, params):
        want.update(params={k: v.clone() for k, v in params.items()}, obs_rms=obs_rms, lr=lr, mu=data['mu'].clone())

    run_oracle_training(ospec, dict(N=N, T=T, mb=mb, mini_epochs=mini_epochs, epochs=epochs, optim='AdamW'), on_epoch)
    assert agent.last_lr == want['lr']
    rms = agent.obs_rms['obs']
    assert rms.running_mean.shape == (37,) and rms.state_dict()['running_var'].shape == (37,)
    assert_close(cpu(rms.running_mean), want['obs_rms']['running_mean'], 1e-6, 2e-7)
    assert_close(cpu(rms.running_var), want['obs_rms']['running_var'], 1e-5, 2e-7)
    assert_close(cpu(agent.storage.storage_dict['mu'].view(-1, 8)), want['mu'], rtol=1e-4, atol=2e-5)
    for n, p in agent.model.named_parameters():
        assert_close(cpu(p), want['params'][n], rtol=2e-3, atol=2e-5, what=n)
    pad = agent.model.flat[: 128 * 40].view(128, 40)[:, 37:]
    assert float(pad.abs().max()) == 0.0, 'weights of the padding columns stay zero'


@pytest.mark.parametrize('net,N,T,mb', [('full', 1024, 32, 16384), ('full_silu', 1024, 32, 16384), ('mid_lnrelu', 64, 8, 128), ('mid_tanh', 64, 8, 128)])
def test_end_to_end_training_above_8192_rows_vs_oracle(hip, tmp_path, net, N, T, mb):
    """One PPO.train() epoch of the product against the oracle's on the deterministic env with minibatches of 16,384 rows: the
    regime of the per-layer kernels (split-bf16 GEMMs, streamed LayerNorm backward, any-order weight-gradient launches), which
    the reference's own fixtures (64 envs) do not reach.  DFlexAntPPO network, 1,024 envs x 32 steps, 2 mini-epochs.  Also the
    tower kinds that have no reference fixture (no norm + SiLU at 16,384 rows, LayerNorm + ReLU and no norm + Tanh at 128):
    rollout, GAE, statistics, losses, KL-adaptive rate and optimizer against the oracle's whole epoch."""
    from mineral_amd.agents import PPO
    from test_oracle_golden import run_oracle_training
    spec = NETS[net]
    has_norm = spec.get('norm', 'LayerNorm') is not None
    ospec = dict(obs=spec['obs'], act=spec['act'], units=spec['units'], norm=spec.get('norm', 'LayerNorm'), act_type=spec.get('actf', 'SiLU'), fixed_sigma=False,
                 dist_type='dreamerv3_normal')
    mini_epochs, epochs = 2, 1
    cfg = _agent_cfg(dict(spec, fixed=False), N, T, mb, mini_epochs, epochs, 'AdamW', str(tmp_path))
    env = synth.GoldenEnv(N, spec['obs'], spec['act'], device=DEV)
    agent = PPO(cfg, logdir=str(tmp_path), accelerator=None, datasets=None, env=env)
    agent.model.load_state_dict(synth.make_params(synth.actor_critic_shapes(spec['obs'], spec['act'], spec['units'], has_norm, True, False), 5))
    agent.noise_fn = lambda k, shape: synth.gaussianish(shape, 900 + k).to(DEV)
    agent.save = lambda *a, **k: None
    agent._checkpoint_save = lambda *a, **k: None
    got = {}
    orig = agent.train_epoch

    def spy():
        res = orig()
        for k in ('loss/total', 'loss/actor', 'loss/critic', 'explained_var', 'grad_norm/all', 'kl'):
            got[k] = cpu(res[k]).clone()
        return res

    agent.train_epoch = spy
    agent.train()
    want = {}

    def on_epoch(e, pre, data, res, lr, obs_rms, value_rms, params):
        want.update(params={k: v.clone() for k, v in params.items()}, obs_rms=obs_rms, lr=lr, mu=data['mu'].clone(), res=res)

    run_oracle_training(ospec, dict(N=N, T=T, mb=mb, mini_epochs=mini_epochs, epochs=epochs, optim='AdamW'), on_epoch)
    assert agent.last_lr == want['lr'], 'KL-adaptive learning rate path'
    rms = agent.obs_rms['obs']
    assert_close(cpu(rms.running_mean), want['obs_rms']['running_mean'], 1e-6, 2e-7)
    assert_close(cpu(rms.running_var), want['obs_rms']['running_var'], 1e-5, 2e-7)
    assert_close(cpu(agent.storage.storage_dict['mu'].view(-1, spec['act'])), want['mu'], rtol=1e-4, atol=2e-5)
    for k in ('loss/total', 'loss/actor', 'loss/critic', 'explained_var', 'grad_norm/all', 'kl'):
        assert_close(got[k], torch.tensor(np.array(want['res'][k], dtype=np.float32)), rtol=2e-3, atol=2e-5, what=k)
    for n, p in agent.model.named_parameters():
        assert_close(cpu(p), want['params'][n], rtol=2e-3, atol=2e-5, what=n)


def test_full_size_properties(hip):
    """BASELINE-size shapes (N=65536 envs, T=32, minibatch 32768) through size-independent properties."""
    from mineral_amd.common.normalizers import RunningMeanStd
    spec = NETS['full']
    N, T, mb = 65536, 32, 32768
    buf = make_buffer(spec, N, T, mb)
    g = torch.Generator(device=DEV); g.manual_seed(1)
    s = buf.storage_dict
    s['rewards'].normal_(generator=g); s['values'].normal_(generator=g)
    s['dones'].copy_((torch.rand(T, N, device=DEV, generator=g) < 0.01).to(torch.uint8))
    s['neglogp'].copy_(torch.arange(T * N, device=DEV, dtype=torch.float32).view(T, N) % 4096)
    last = torch.randn(N, device=DEV, generator=g)
    vr = RunningMeanStd((1,), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    buf.compute_return(last, 0.99, 0.95, vr)
    d = buf.prepare_training()
    torch.cuda.synchronize()
    # index map: neglogp payload is a function of (t, n)
    idx = torch.arange(N * T, device=DEV)
    n, t = idx // T, idx % T
    assert torch.equal(d['neglogp'], ((t * N + n) % 4096).float())
    adv = d['advantages'].double()
    assert abs(adv.mean().item()) < 1e-5 and abs(adv.std().item() - 1.0) < 1e-5
    assert vr.running_count.item() == 1 + 2 * N * T
    # terminal rows: with done[t]=1 the return of step t is reward + 0 (no bootstrap), checked on the raw recursion
    buf2 = make_buffer(spec, N, T, mb)
    for k in ('rewards', 'values', 'dones', 'neglogp'):
        buf2.storage_dict[k].copy_(s[k])
    buf2.compute_return(last, 0.99, 0.95, None)
    d2 = buf2.prepare_training()
    ret_tn = d2['returns'].view(N, T).t()
    done = s['dones'].bool()
    r_, v_ = s['rewards'].squeeze(-1)[done], s['values'].squeeze(-1)[done]
    assert torch.equal(ret_tn[done], (r_ - v_) + v_), 'a terminal step returns (r - V) + V: no bootstrap, no carried advantage'
    # full-size minibatch gradient: finite everywhere and bit-reproducible (fixed-order reductions, no atomics)
    params = oracle_params(spec, 4)
    model = make_model(spec, params)
    s['obses']['obs'].normal_(generator=g)
    s['actions'].normal_(generator=g).mul_(0.5)
    L = hip.lib()
    ws = torch.empty(int(L.mrl_workspace_bytes(model.net, mb)), dtype=torch.uint8, device=DEV)
    chp = hip.PpoHp(0.2, 4.0, 0.0, 1e-4, 0, 0, 0)
    mean_f = torch.zeros(64, device=DEV); std_f = torch.ones(64, device=DEV)
    metrics = torch.zeros(hip.NMETRIC, dtype=torch.float64, device=DEV)
    outs = []
    for rep in range(2):
        s['mu'].zero_(); s['sigma'].fill_(0.6)
        buf._flat['neglogp'].fill_(12.0)
        gr = torch.full_like(model.flat, float('nan'))
        hip.check(L.mrl_ppo_grad(model.net, hip.ptr(model.flat), chp, buf.c, mb, mb, hip.ptr(mean_f), hip.ptr(std_f), -5.0, 5.0,
                                 hip.ptr(gr), hip.ptr(metrics), hip.ptr(ws), ws.numel(), None), 'grad')
        outs.append(gr)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), 'gradients are bit-reproducible'
    assert torch.isfinite(metrics).all()


@pytest.mark.gpu
@pytest.mark.parametrize('mb', [4096, 16384])
def test_two_ranks_share_one_gpu_through_host_staged_gloo(mb):
    """The multi-rank agent path (parameter broadcast, sharded GAE statistics, merged observation moments, gradient
    all-reduce, KL all-reduce) run as two processes on this GPU with gloo through host memory: both ranks must end
    with bit-identical parameters, with the kernels of both minibatch regimes (chain kernels up to 8192 rows, per-layer
    split-bf16 kernels above).  The ranks step different environments, so there is no single-process run with the same
    numbers to compare with; that comparison is made with one rank (test_one_rank_rccl_group_matches_the_single_process_path).
    RCCL itself needs one GPU per rank and is exercised by the scaling benchmark."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--backend', 'gloo',
           '--num-envs', '1024', '--minibatch', str(mb), '--no-kernel-timing']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['ranks_in_sync'] is True
    assert out['rccl'] == {'backend': 'gloo', 'world_size': 2, 'ranks_seen': [0, 1], 'devices_seen': [0, 0]}
    assert out['value'] > 0


@pytest.mark.gpu
@pytest.mark.parametrize('N,L', [(37, 10), (4096, 100), (70001, 100), (5, 8)])
def test_device_metrics_window_equals_deque_semantics(N, L):
    """mineral/common/metrics.py:59-67 + tracker.py through `mrl_episode_update`: window of the last L finished episodes, env
    order, counts and running scores, against a deque on the host."""
    import collections
    from mineral_amd.common.metrics import Metrics
    scores = {'rewards': torch.zeros(N, device=DEV), 'lengths': torch.zeros(N, dtype=torch.int64, device=DEV)}
    m = Metrics(scores, L)
    ref_r, ref_l = collections.deque([0] * L, maxlen=L), collections.deque([0] * L, maxlen=L)
    cur_r, cur_l = np.zeros(N, dtype=np.float32), np.zeros(N, dtype=np.int64)
    rng = np.random.RandomState(0)
    total = 0
    for step in range(40):
        r = rng.randn(N).astype(np.float32)
        p = 0.9 if step == 17 else (0.0 if step == 5 else 0.15)   # one step finishes more episodes than the window holds, one none
        d = rng.rand(N) < p
        dt = torch.from_numpy(d).to(DEV)
        if step % 3 == 1:
            dt = dt.float()                                        # envs may hand over float / integer done flags
        m.update(0, None, None, torch.from_numpy(r).to(DEV), dt)
        cur_r = (cur_r + r).astype(np.float32)
        cur_l += 1
        idx = np.nonzero(d)[0]
        ref_r.extend(cur_r[idx].tolist())
        ref_l.extend(cur_l[idx].tolist())
        cur_r[idx] = 0
        cur_l[idx] = 0
        total += len(idx)
    assert m.num_episodes == total
    assert m.episode_trackers['rewards'].window == [float(x) for x in ref_r]          # same fp32 running sums, bit for bit
    assert [int(x) for x in m.episode_trackers['lengths'].window] == list(ref_l)
    assert np.array_equal(scores['rewards'].cpu().numpy(), cur_r) and np.array_equal(scores['lengths'].cpu().numpy(), cur_l)
    assert abs(m.episode_trackers['rewards'].mean() - float(np.mean(ref_r))) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('case', sorted(synth.METRICS_CASES))
def test_info_key_metrics_vs_reference(golden, case):
    """Metrics with regex-selected info keys (metrics.py:69-113: sum / min / max / final of per-env scalars of the observation
    dict, appended to a window when the env finishes) on the device against the windows and result() recorded from the
    reference class (tests/golden/make_golden_metrics.py)."""
    from mineral_amd.common.metrics import Metrics
    g = golden('g17_metrics_info_keys')
    N, L, steps, _ = synth.METRICS_CASES[case]
    scores = {'rewards': torch.zeros(N, device=DEV), 'lengths': torch.zeros(N, dtype=torch.int64, device=DEV)}
    m = Metrics(scores, L, info_keys=synth.METRICS_INFO_KEYS)
    for t in range(steps):
        obs, r, d = synth.metrics_step(case, t)
        m.update(0, None, {k: v.to(DEV) for k, v in obs.items()}, r.to(DEV), d.to(DEV), {})
    assert m.num_episodes == int(g[f'{case}/num_episodes'])
    names = [k.split('/', 2)[2] for k in g if k.startswith(f'{case}/window/')]
    assert sorted(names) == sorted(m.episode_trackers), (names, list(m.episode_trackers))
    for k in names:
        # fp32 running sums / extrema of the same values in the same order: bit-equal
        assert m.episode_trackers[k].window == g[f'{case}/window/{k}'].tolist(), k
    res = m.result(prefix='train')
    want = {k.split('/', 2)[2]: float(v) for k, v in g.items() if k.startswith(f'{case}/result/')}
    assert sorted(res) == sorted(want)
    for k, v in want.items():
        assert abs(res[k] - v) <= 1e-12 * max(1.0, abs(v)), k


@pytest.mark.gpu
def test_metrics_refuse_what_the_device_path_does_not_aggregate():
    from mineral_amd.common.metrics import Metrics
    scores = {'rewards': torch.zeros(4, device=DEV), 'lengths': torch.zeros(4, dtype=torch.int64, device=DEV)}
    with pytest.raises(NotImplementedError, match='video'):
        Metrics(scores, 10, save_video_every=5)
    with pytest.raises(NotImplementedError, match='video'):
        Metrics(scores, 10, info_keys={'video': 'rgb'})
    m = Metrics(scores, 10, info_keys={'sum': 'obs'})
    with pytest.raises(NotImplementedError, match='one value per env'):
        m.update(0, None, {'obs': torch.zeros(4, 8, device=DEV)}, torch.zeros(4, device=DEV), torch.zeros(4, dtype=torch.bool, device=DEV))


@pytest.mark.gpu
@pytest.mark.parametrize('tag,net,dist', [('full_dv3', 'full', 'dreamerv3_normal'), ('cartpole', 'cartpole', 'normal')])
def test_actor_critic_act_and_forward_api(hip, golden, tag, net, dist):
    """ActorCritic.act / forward (models.py:93-126) of the parameter container against the reference module's own outputs
    (golden g4: act with injected noise, forward with prev_actions), and act(sample=False) against the oracle's raw mu head."""
    g = golden(f'g4_model_{tag}')
    spec = NETS[net]
    params = synth.make_params(synth.actor_critic_shapes(spec['obs'], spec['act'], spec['units'], spec.get('norm', 'LayerNorm') is not None, True,
                                                         spec.get('fixed', False)), 3)      # the parameters the fixture was recorded with
    model = make_model(spec, params, dist)
    obs, prev, noise = synth.gaussianish((16, spec['obs']), 41), synth.gaussianish((16, spec['act']), 42), synth.gaussianish((16, spec['act']), 43)
    tol = dict(rtol=2e-5, atol=2e-5)
    act = model.act({'obs': dev(obs)}, noise=dev(noise))
    for k in ('actions', 'mu', 'sigma', 'values'):
        assert_close(cpu(act[k]), g[f'act_{k}'], **tol, what=f'act {k}')
    assert_close(cpu(act['neglogp']), g['act_neglogp'], rtol=2e-5, atol=1e-4, what='act neglogp')
    assert act['values'].shape == (16, 1) and isinstance(act['distr'], torch.distributions.Normal)
    out = model({'obs': dev(obs), 'prev_actions': dev(prev)})
    for k in ('values', 'mu', 'sigma'):
        assert_close(cpu(out[k]), g[k], **tol, what=f'forward {k}')
    assert_close(cpu(out['entropy']), g['entropy'], rtol=2e-5, atol=1e-4, what='entropy')
    assert_close(cpu(out['prev_neglogp']), g['prev_neglogp'], rtol=5e-5, atol=2e-4, what='prev_neglogp')
    raw = model.act({'obs': dev(obs)}, sample=False)
    assert isinstance(raw, torch.Tensor) and raw.shape == (16, spec['act'])
    mu_raw, _, _ = O.actor_critic_raw(params, obs, oracle_cfg(spec, dist))
    assert_close(cpu(raw), mu_raw, **tol, what='act(sample=False): mu before the distribution transform')


@pytest.mark.gpu
def test_torch_custom_operators_run_the_same_kernels(hip):
    """torch.ops.mineral.* (mineral_amd/ops.py: torch.library registrations over the C ABI) against the direct calls: GAE +
    normalisation, one minibatch gradient, clip + AdamW, the running-statistics merge and normalize: identical bits."""
    import mineral_amd.ops  # noqa: F401  (registers the operators)
    from mineral_amd.common.normalizers import RunningMeanStd
    M = torch.ops.mineral
    spec = NETS['mid']
    N, T, mb = 64, 8, 256
    net_i, net_f = [spec['obs'], spec['act'], 1, 0, 0] + spec['units'], [0.1, 1.0, 1e-5]
    params = oracle_params(spec, 9)
    model = make_model(spec, params)
    assert M.param_count(net_i, net_f) == model.param_count
    L = hip.lib()

    def rollout_args(buf):
        s = buf.storage_dict
        flat = torch.stack([buf._flat[k] for k in ('values', 'returns', 'advantages', 'neglogp')])
        return (buf.obs_store, s['actions'], s['mu'], s['sigma'], s['neglogp'], s['values'], s['rewards'], s['dones'], flat), flat

    # --- GAE + value normalisation
    bufs = [make_buffer(spec, N, T, mb) for _ in range(2)]
    for b in bufs:
        s = b.storage_dict
        s['rewards'].copy_(synth.gaussianish((T, N, 1), 11)); s['values'].copy_(synth.gaussianish((T, N, 1), 12))
        s['dones'].copy_((synth.lcg_t((T, N), 14) > 0.8).to(torch.uint8)); s['neglogp'].copy_(synth.gaussianish((T, N), 13))
    last = dev(synth.gaussianish((N,), 18))
    vr = [RunningMeanStd((1,), eps=1e-5, with_clamp=True, initial_count=1, device=DEV) for _ in range(2)]
    bufs[0].compute_return(last, 0.99, 0.95, vr[0])
    args, flat = rollout_args(bufs[1])
    stats = torch.zeros(8, device=DEV)
    ws_g = torch.empty(8192, dtype=torch.uint8, device=DEV)
    M.gae_finish(*args, last, 0.99, 0.95, vr[1].running_mean, vr[1].running_var, vr[1].running_count, 1e-5, -5.0, 5.0, stats, ws_g)
    for i, k in enumerate(('values', 'returns', 'advantages', 'neglogp')):
        assert torch.equal(flat[i], bufs[0]._flat[k]), k
    assert torch.equal(vr[0].running_var, vr[1].running_var) and torch.equal(vr[0].running_count, vr[1].running_count)

    # --- one minibatch gradient + clip + AdamW
    outs = []
    for use_ops in (False, True):
        buf = make_buffer(spec, N, T, mb)
        _fill_training_buffer(spec, buf, N, T)
        m2 = make_model(spec, params)
        ws = torch.empty(int(M.workspace_bytes(net_i, net_f, mb)), dtype=torch.uint8, device=DEV)
        grads = torch.full((m2.param_count,), float('nan'), device=DEV)
        metrics = torch.zeros(hip.NMETRIC, dtype=torch.float64, device=DEV)
        m1, mv, gn = torch.zeros_like(m2.flat), torch.zeros_like(m2.flat), torch.zeros(1, device=DEV)
        scratch = torch.empty(4096, dtype=torch.uint8, device=DEV)
        mean_f, std_f = torch.zeros(spec['obs'], device=DEV), torch.ones(spec['obs'], device=DEV)
        if use_ops:
            args, _ = rollout_args(buf)
            M.ppo_grad(net_i, net_f, m2.flat, [0.2, 4.0, 0.01, 1e-4], [0, 0, 0], *args, mb, mb, mean_f, std_f, -5.0, 5.0, grads, metrics, ws)
            M.clip_adam_step_(m2.flat, grads, m1, mv, 5e-4, 0.9, 0.999, 1e-5, 0.01, True, 1, 0.5, 1.0, gn, scratch)
        else:
            chp = hip.PpoHp(0.2, 4.0, 0.01, 1e-4, 0, 0, 0)
            hip.check(L.mrl_ppo_grad(m2.net, hip.ptr(m2.flat), chp, buf.c, mb, mb, hip.ptr(mean_f), hip.ptr(std_f), -5.0, 5.0, hip.ptr(grads),
                                     hip.ptr(metrics), hip.ptr(ws), ws.numel(), None), 'grad')
            hip.check(L.mrl_adam_step(hip.ptr(m2.flat), hip.ptr(grads), hip.ptr(m1), hip.ptr(mv), m2.param_count, 5e-4, 0.9, 0.999, 1e-5, 0.01, 1, 1, 0.5,
                                      1.0, hip.ptr(gn), hip.ptr(scratch), None), 'adam')
        torch.cuda.synchronize()
        outs.append((grads.clone(), m2.flat.clone(), metrics.clone(), gn.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[0][0]).all()

    # --- running statistics
    r = RunningMeanStd((6,), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    x = dev(synth.gaussianish((50, 6), 100) * 2 + 0.3)
    want_n = None
    bm, bv = x.mean(0), x.var(0, correction=1)
    mean, var, cnt = r.running_mean.clone(), r.running_var.clone(), r.running_count.clone()
    r.update(x)
    M.welford_update_(mean, var, cnt, 1e-5, bm, bv, 50.0, None, None)
    assert torch.allclose(mean, r.running_mean, rtol=1e-6, atol=1e-9) and cnt.item() == r.running_count.item()
    y = M.normalize_clamp(x * 3, r.running_mean, r.running_var, r.running_count, 1e-5, -5.0, 5.0, False)
    assert torch.equal(y, r.normalize(x * 3))
    with pytest.raises(NotImplementedError):
        M.normalize_clamp(x.cpu(), r.running_mean.cpu(), r.running_var.cpu(), r.running_count.cpu(), 1e-5, -5.0, 5.0, False)


@pytest.mark.gpu
@pytest.mark.parametrize('mb', [4096, 16384])
def test_one_rank_rccl_group_matches_the_single_process_path(mb):
    """The same multi-rank code path over RCCL (backend 'nccl') with a one-rank group on this GPU: every collective the agent
    issues runs through RCCL on device tensors, and the parameters after an update must equal the single-process run's
    bit for bit (global statistics of one rank are its local ones).  The single-process run takes the one-call mini-epoch; the
    ranked run the same call with the all-reduce hook (mrl_ppo_mini_epoch_reduce), and with MRL_RANKED_CALLS=1 the merge /
    gradient / all-reduce / optimizer calls per minibatch; 4096 rows run the chain kernels, 16384 the per-layer split-bf16
    kernels."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra, calls in (([], '0'), (['--force-dist'], '0'), (['--force-dist'], '1')):
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--steps', '1', '--warmup', '1', '--num-envs', '1024', '--minibatch', str(mb),
               '--no-kernel-timing', '--no-cpu-baseline', '--no-secondary'] + extra
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), MRL_RANKED_CALLS=calls)
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1]))
    for o in outs[1:]:
        assert o['ranks_in_sync'] is True
        assert outs[0]['param_checksum'] == o['param_checksum']
        assert outs[0]['param_sha256'] == o['param_sha256']
        assert o['collectives_per_update']['all_reduce'] == outs[1]['collectives_per_update']['all_reduce']


@pytest.mark.gpu
@pytest.mark.parametrize('mb', [256, 16384])
def test_an_optimizer_step_replays_from_a_hip_graph(hip, mb):
    """Callers may capture mrl_rms_merge -> mrl_ppo_grad -> mrl_adam_step into a HIP graph (the library allocates nothing and
    records no memset; above 8192 rows it issues any-order launches).  Every replay, with eager work and synchronisation
    between replays, must leave exactly the state the eager calls leave from the same starting point."""
    from mineral_amd.common.normalizers import RunningMeanStd
    spec = NETS['full']
    N, T = mb // 8, 8
    params = oracle_params(spec, 9)
    model = make_model(spec, params)
    buf = make_buffer(spec, N, T, mb)
    _fill_training_buffer(spec, buf, N, T)
    L = hip.lib()
    P = model.param_count
    ws = torch.empty(int(L.mrl_workspace_bytes(model.net, mb)), dtype=torch.uint8, device=DEV)
    grads, m, v = (torch.zeros(P, device=DEV) for _ in range(3))
    metrics = torch.zeros(hip.NMETRIC, dtype=torch.float64, device=DEV)
    gnorm = torch.zeros(1, device=DEV)
    scratch = torch.empty(4096, dtype=torch.uint8, device=DEV)
    rms = RunningMeanStd((spec['obs'],), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    obs2d = buf.storage_dict['obses']['obs'].reshape(N * T, -1)
    bmean, bvar = obs2d.mean(0).contiguous(), obs2d.var(0, correction=1).contiguous()
    mean_f, std_f = torch.empty(spec['obs'], device=DEV), torch.empty(spec['obs'], device=DEV)
    chp = hip.PpoHp(0.2, 4.0, 0.0, 1e-4, 0, 0, 0)
    c = rms.c_struct()

    def step(s):
        hip.check(L.mrl_rms_merge(c, hip.ptr(bmean), hip.ptr(bvar), float(mb), hip.ptr(mean_f), hip.ptr(std_f), s), 'merge')
        hip.check(L.mrl_ppo_grad(model.net, hip.ptr(model.flat), chp, buf.c, 0, mb, hip.ptr(mean_f), hip.ptr(std_f), -5.0, 5.0, hip.ptr(grads),
                                 hip.ptr(metrics), hip.ptr(ws), ws.numel(), s), 'grad')
        hip.check(L.mrl_adam_step(hip.ptr(model.flat), hip.ptr(grads), hip.ptr(m), hip.ptr(v), P, 5e-4, 0.9, 0.999, 1e-5, 0.01, 1, 3, 0.5, 1.0,
                                  hip.ptr(gnorm), hip.ptr(scratch), s), 'adam')

    state = [model.flat, m, v, rms.running_mean, rms.running_var, rms.running_count, buf.storage_dict['mu'], buf.storage_dict['sigma']]
    start = [t.clone() for t in state]

    def restore():
        for t, t0 in zip(state, start):
            t.copy_(t0)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step(side.cuda_stream)                       # eager, also the warm-up before the capture
    side.synchronize()
    want = [t.clone() for t in state] + [metrics.clone(), gnorm.clone()]
    assert all(bool(torch.isfinite(t).all()) for t in want)
    restore()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        restore()
        metrics.zero_(); gnorm.zero_(); grads.fill_(float('nan'))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        got = state + [metrics, gnorm]
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), f'replay {rep}: tensor {i} differs from the eager step'


@pytest.mark.gpu
@pytest.mark.parametrize('mb,net', [(256, 'full'), (2048, 'full'), (296, 'cartpole'), (16384, 'full')])
def test_the_one_call_step_and_mini_epoch_equal_the_three_calls(hip, mb, net):
    """mrl_ppo_step folds the launches of mrl_rms_merge -> mrl_ppo_grad -> mrl_adam_step into each other (statistics merge in the
    first launch, gradient norm taken while the gradient is written), and mrl_ppo_mini_epoch runs the steps of consecutive
    minibatches in one call (the optimizer writes the weights straight into the layouts the next step reads).  Parameters, Adam
    moments, running statistics, metrics, gradient and gradient norm after three steps on three minibatches must equal the
    three calls' bit for bit."""
    from mineral_amd.common.normalizers import RunningMeanStd
    spec = NETS[net]
    T, steps = 8, 3
    N = steps * mb // T
    params = oracle_params(spec, 9)
    L = hip.lib()
    chp = hip.PpoHp(0.2, 4.0, 0.01, 1e-4, 0, 0, 0)
    outs = []
    for mode in ('three calls', 'one call per step', 'one call'):
        model = make_model(spec, params)
        buf = make_buffer(spec, N, T, mb)
        _fill_training_buffer(spec, buf, N, T)
        P = model.param_count
        ws = torch.zeros(int(L.mrl_workspace_bytes(model.net, mb)), dtype=torch.uint8, device=DEV)
        grads, m, v = (torch.zeros(P, device=DEV) for _ in range(3))
        metrics = torch.zeros(steps, hip.NMETRIC, dtype=torch.float64, device=DEV)
        gnorm = torch.zeros(steps, device=DEV)
        scratch = torch.empty(4096, dtype=torch.uint8, device=DEV)
        rms = RunningMeanStd((spec['obs'],), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
        obs3d = buf.storage_dict['obses']['obs'].reshape(steps, mb, -1)
        bmean, bvar = obs3d.mean(1).contiguous(), obs3d.var(1, correction=1).contiguous()
        mean_f, std_f = torch.empty(spec['obs'], device=DEV), torch.empty(spec['obs'], device=DEV)
        c = rms.c_struct()
        if mode == 'one call':
            hip.check(L.mrl_ppo_mini_epoch(model.net, hip.ptr(model.flat), chp, buf.c, mb, steps, c, hip.ptr(bmean), hip.ptr(bvar), float(mb), hip.ptr(mean_f),
                                           hip.ptr(std_f), -5.0, 5.0, hip.ptr(grads), hip.ptr(metrics), hip.ptr(ws), ws.numel(), hip.ptr(m), hip.ptr(v),
                                           5e-4, 0.9, 0.999, 1e-5, 0.01, 1, 1, 0.5, hip.ptr(gnorm), hip.ptr(scratch), None), 'mini-epoch')
        for k in range(steps if mode != 'one call' else 0):
            if mode == 'one call per step':
                hip.check(L.mrl_ppo_step(model.net, hip.ptr(model.flat), chp, buf.c, k * mb, mb, c, hip.ptr(bmean[k]), hip.ptr(bvar[k]), float(mb),
                                         hip.ptr(mean_f), hip.ptr(std_f), -5.0, 5.0, hip.ptr(grads), hip.ptr(metrics[k]), hip.ptr(ws), ws.numel(), hip.ptr(m),
                                         hip.ptr(v), 5e-4, 0.9, 0.999, 1e-5, 0.01, 1, k + 1, 0.5, hip.ptr(gnorm[k:]), hip.ptr(scratch), None), 'step')
            else:
                hip.check(L.mrl_rms_merge(c, hip.ptr(bmean[k]), hip.ptr(bvar[k]), float(mb), hip.ptr(mean_f), hip.ptr(std_f), None), 'merge')
                hip.check(L.mrl_ppo_grad(model.net, hip.ptr(model.flat), chp, buf.c, k * mb, mb, hip.ptr(mean_f), hip.ptr(std_f), -5.0, 5.0, hip.ptr(grads),
                                         hip.ptr(metrics[k]), hip.ptr(ws), ws.numel(), None), 'grad')
                hip.check(L.mrl_adam_step(hip.ptr(model.flat), hip.ptr(grads), hip.ptr(m), hip.ptr(v), P, 5e-4, 0.9, 0.999, 1e-5, 0.01, 1, k + 1, 0.5, 1.0,
                                          hip.ptr(gnorm[k:]), hip.ptr(scratch), None), 'adam')
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (model.flat, m, v, grads, metrics, gnorm, rms.running_mean, rms.running_var, rms.running_count, mean_f, std_f,
                                         buf.storage_dict['mu'], buf.storage_dict['sigma'])])
    assert all(bool(torch.isfinite(t).all()) for t in outs[0])
    assert float(outs[0][5].min()) > 0.0
    for o, mode in zip(outs[1:], ('one call per step', 'one call')):
        for i, (a, b) in enumerate(zip(outs[0], o)):
            assert torch.equal(a, b), f'tensor {i} differs between the three calls and {mode}'


@pytest.mark.gpu
def test_launch_structures_of_the_update_are_bit_identical():
    """On the per-layer path (batches above 8192 rows) mrl_ppo_grad lets the weight-gradient launches run beside the dgrad launch
    that reads the same dZ (any-order launches on the caller's stream; MRL_ANY_ORDER=0 keeps every launch in order), or, with
    MRL_TOWER_STREAMS=1, runs one launch chain per tower plus the weight gradients on library-owned streams (fork / join by
    events).  Same kernels, same fixed-order reductions: every bit of every parameter after two updates (sha256 of the vector)
    must be the same in all three."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = []
    for env in (dict(MRL_ANY_ORDER='0', MRL_TOWER_STREAMS='0'), dict(MRL_ANY_ORDER='1', MRL_TOWER_STREAMS='0'), dict(MRL_TOWER_STREAMS='1')):
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--steps', '1', '--warmup', '1', '--num-envs', '1024', '--minibatch', '16384',
               '--no-kernel-timing', '--no-cpu-baseline', '--no-secondary']
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, **env))
        assert res.returncode == 0, res.stderr[-2000:]
        sums.append(json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])['param_sha256'])
    assert sums[0] == sums[1] == sums[2]


@pytest.mark.gpu
def test_launcher_cli_trains_and_writes_the_reference_artifacts(tmp_path):
    """`python -m mineral.scripts.run task=Synthetic ...` (the drop-in launch line) trains for a few epochs on the HIP path
    and leaves the on-disk artefacts of mineral/scripts/run.py:34-41 and ppo.py:416-426."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logdir = str(tmp_path / 'run')
    cmd = [sys.executable, '-m', 'mineral.scripts.run', 'task=Synthetic', 'num_envs=256', 'agent.ppo.minibatch_size=2048',
           'agent.ppo.max_agent_steps=24576', 'agent.print_every=1', f'logdir={logdir}', 'seed=3']
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'Traceback' not in res.stderr and 'Exception:' not in res.stdout, (res.stdout[-1500:], res.stderr[-1500:])
    for rel in ('run_metadata.yaml', 'resolved_config.yaml', 'ckpt/final.pth'):
        assert os.path.exists(os.path.join(logdir, rel)), rel
    ckpt = torch.load(os.path.join(logdir, 'ckpt', 'final.pth'), map_location='cpu', weights_only=True)
    assert {'epoch', 'mini_epoch', 'agent_steps', 'model', 'optim', 'obs_rms', 'value_rms'} <= set(ckpt)
    assert ckpt['agent_steps'] == 24576 and ckpt['epoch'] == 2
    assert 'actor_mlp.mlp.0.weight' in ckpt['model'] and ckpt['obs_rms']['obs.running_count'].dtype == torch.float64


@pytest.mark.gpu
def test_launcher_cli_runs_the_cartpole_network(tmp_path):
    """`agent=CartpolePPO` (no normalisation layer, ELU, state-independent log-std; mineral/cfgs/agent/CartpolePPO.yaml) through
    the launcher: the checkpoint carries the reference's parameter names for that network."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logdir = str(tmp_path / 'run')
    cmd = [sys.executable, '-m', 'mineral.scripts.run', 'task=Synthetic', 'agent=CartpolePPO', 'num_envs=128', 'agent.ppo.minibatch_size=1024',
           'agent.ppo.max_agent_steps=6144', 'agent.print_every=1', f'logdir={logdir}', 'seed=3']
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'Traceback' not in res.stderr and 'Exception:' not in res.stdout, (res.stdout[-1500:], res.stderr[-1500:])
    ckpt = torch.load(os.path.join(logdir, 'ckpt', 'final.pth'), map_location='cpu', weights_only=True)
    names = list(ckpt['model'])
    assert names[0] == 'sigma' and ckpt['model']['sigma'].shape == (12,)
    assert 'actor_mlp.mlp.2.weight' in names and 'actor_mlp.mlp.1.weight' not in names and 'sigma.weight' not in names
    assert ckpt['model']['actor_mlp.mlp.0.weight'].shape == (32, 64) and ckpt['agent_steps'] == 6144
    assert all(torch.isfinite(v).all() for v in ckpt['model'].values())


@pytest.mark.gpu
def test_launcher_cli_runs_the_dflex_ant_config_at_four_envs(tmp_path):
    """The reference's smallest case -- `agent=DFlexAntPPO` with num_envs=4 -- on DFlex Ant's shapes (37 observations, 8 actions)
    through the launcher.  The simulator is not in the image, so the synthetic env supplies tensors of those shapes; the
    reference runs this case on the CPU, the product on the device (it has no CPU path).  minibatch_size is the whole
    4 x 32 batch, as the reference's own assert (ppo.py:83) demands of any override with so few envs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logdir = str(tmp_path / 'run')
    cmd = [sys.executable, '-m', 'mineral.scripts.run', 'task=Synthetic', 'agent=DFlexAntPPO', 'num_envs=4', 'task.env.obs_dim=37',
           'task.env.act_dim=8', 'agent.ppo.minibatch_size=128', 'agent.ppo.max_agent_steps=512', 'agent.print_every=1', f'logdir={logdir}', 'seed=7']
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'Traceback' not in res.stderr and 'Exception:' not in res.stdout, (res.stdout[-1500:], res.stderr[-1500:])
    ckpt = torch.load(os.path.join(logdir, 'ckpt', 'final.pth'), map_location='cpu', weights_only=True)
    assert ckpt['agent_steps'] == 512 and ckpt['epoch'] == 3      # four updates; the counter starts at -1 (agent.py:79)
    assert ckpt['model']['actor_mlp.mlp.0.weight'].shape == (512, 37) and ckpt['model']['mu.weight'].shape == (8, 128)
    assert ckpt['obs_rms']['obs.running_mean'].shape == (37,)
    assert all(torch.isfinite(v).all() for v in ckpt['model'].values())


@pytest.mark.gpu
def test_env_autoresets_false_resets_finished_envs_before_the_next_step(hip, tmp_path):
    """task.env_autoresets=False (ppo.py:136-142): finished envs are reset through env.reset_idx and their fresh observation is
    what the next policy step sees and stores."""
    from mineral_amd.agents import PPO

    class ManualResetEnv(synth.GoldenEnv):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.reset_calls = []

        def reset_idx(self, ids):
            self.reset_calls.append((self.t, list(ids)))
            return torch.full((len(ids), self.obs_dim), 7.0 + self.t, device=self.device)

    spec = NETS['mid']
    N, T = 16, 8
    cfg = _agent_cfg(spec, N, T, N * T, 1, 1, 'Adam', str(tmp_path))
    cfg.task['env_autoresets'] = False
    env = ManualResetEnv(N, spec['obs'], spec['act'], device=DEV)       # episode_length 5, staggered counters
    agent = PPO(cfg, logdir=str(tmp_path), accelerator=None, datasets=None, env=env)
    agent.save = lambda *a, **k: None
    agent._checkpoint_save = lambda *a, **k: None
    agent.noise_fn = lambda k, shape: synth.gaussianish(shape, 900 + k).to(DEV)
    agent.train()
    obs_buf = cpu(agent.storage.storage_dict['obses']['obs'])            # (N, T, O)
    dones = cpu(agent.storage.storage_dict['dones'])                     # (T, N) as stored per step
    assert env.reset_calls, 'some env must have finished within the horizon'
    for t_env, ids in env.reset_calls:
        # the reset happens at the top of rollout step t_env (env.t counts completed steps), after the dones of step t_env - 1
        assert sorted(ids) == torch.nonzero(dones.reshape(T, N)[t_env - 1]).flatten().tolist()
        assert torch.all(obs_buf[ids, t_env] == 7.0 + t_env)
        others = [i for i in range(N) if i not in ids]
        assert not torch.any(obs_buf[others, t_env] == 7.0 + t_env)

