"""HIP SAC update loop through the C ABI: sampling, critic / actor / alpha gradients and Polyak against the oracle's autograd
(fp32; tolerances written at each comparison), then the product agent against the run recorded from the reference."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import synth  # noqa: E402
from oracle import ppo_oracle as P  # noqa: E402
from oracle import sac_oracle as S  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def pack(params, shapes, net, which):
    """Flat device vector in the library's layout from an oracle parameter dict (reference state_dict names)."""
    from mineral_amd import _hip
    L = _hip.lib()
    n = int(L.mrl_sac_param_count(net, which))
    offs = (C.c_int64 * 64)()
    cnt = C.c_int32(0)
    _hip.check(L.mrl_sac_param_offsets(net, which, offs, C.byref(cnt)), 'offsets')
    flat = torch.zeros(n)
    order = layout_names(net, which)
    assert len(order) == cnt.value
    for name, off in zip(order, list(offs)[:cnt.value]):
        t = params[name].reshape(-1)
        flat[off:off + t.numel()] = t
    return flat.to(DEV), dict(zip(order, list(offs)[:cnt.value]))


def layout_names(net, which):
    """Parameter names in the order of mrl_sac_param_offsets."""
    if which == 0:
        names = []
        for l in range(net.actor_layers):
            b = 3 * l
            names += [f'actor_mlp.mlp.{b}.weight', f'actor_mlp.mlp.{b}.bias', f'actor_mlp.mlp.{b + 1}.weight', f'actor_mlp.mlp.{b + 1}.bias']
        return names + ['HEAD.weight', 'HEAD.bias']
    names = []
    for c in range(2):
        for l in range(net.critic_layers):
            b = 3 * l
            names += [f'critics.{c}.mlp.{b}.weight', f'critics.{c}.mlp.{b}.bias', f'critics.{c}.mlp.{b + 1}.weight', f'critics.{c}.mlp.{b + 1}.bias']
    b = 3 * net.critic_layers
    for c in range(2):
        names += [f'critics.{c}.mlp.{b}.weight', f'critics.{c}.mlp.{b}.bias']
    return names


def with_head(pa):
    """The actor's mu / sigma heads stacked like the library's [mu; sigma] head."""
    d = dict(pa)
    d['HEAD.weight'] = torch.cat([pa['mu.weight'], pa['sigma.weight']], 0)
    d['HEAD.bias'] = torch.cat([pa['mu.bias'], pa['sigma.bias']], 0)
    return d


def unpack_grads(flat, offsets, shapes_of):
    out = {}
    f = flat.cpu()
    for name, off in offsets.items():
        n = int(np.prod(shapes_of[name]))
        out[name] = f[off:off + n].reshape(shapes_of[name])
    return out


def rel_err(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


@pytest.mark.parametrize('O,A,au,cu,B', [(8, 4, [64, 32], [64, 32], 48), (64, 12, [512, 256, 128], [512, 256, 128], 300), (12, 4, [128], [256, 64], 129)])
def test_sac_gradients_vs_oracle(O, A, au, cu, B):
    from mineral_amd import _hip
    L = _hip.lib()
    cfg = S.SACConfig(O, A, au, cu)
    hp = S.SACHParams()
    net = _hip.make_sac_net(O, A, au, cu)
    pa = synth.make_params(S.actor_shapes(cfg), 31)
    pc = synth.make_params(S.critic_shapes(cfg), 32)
    pt = synth.make_params(S.critic_shapes(cfg), 33)
    pa['sigma.bias'] = pa['sigma.bias'] + torch.linspace(-7.0, 3.0, A)     # some log-stds outside the clamp
    fa, offs_a = pack(with_head(pa), None, net, 0)
    fc, offs_c = pack(pc, None, net, 1)
    ft, _ = pack(pt, None, net, 1)
    obs, nxt = synth.gaussianish((B, O), 41) * 1.5, synth.gaussianish((B, O), 42) * 1.5
    act = torch.tanh(synth.gaussianish((B, A), 43))
    rew, done = synth.gaussianish((B, 1), 44), (synth.lcg_t((B, 1), 45) > 0.3).float()
    eps_n, eps_a = synth.gaussianish((B, A), 46), synth.gaussianish((B, A), 47)
    mean, var = synth.gaussianish((O,), 48) * 0.3, synth.lcg_t((O,), 49) * 0.5 + 1.0
    rms = {'running_mean': mean, 'running_var': var, 'running_count': torch.tensor(10.0)}
    stdv = torch.sqrt(var + hp.rms_eps)
    log_alpha = torch.tensor(-0.3)
    d = lambda t: t.contiguous().to(DEV)
    d_obs, d_nxt, d_act, d_rew, d_done, d_en, d_ea, d_mean, d_std, d_la = map(d, (obs, nxt, act, rew, done, eps_n, eps_a, mean, stdv, log_alpha.reshape(1)))
    ws = torch.empty(int(L.mrl_sac_workspace_bytes(net, B)), dtype=torch.uint8, device=DEV)
    metrics = torch.zeros(_hip.SAC_NMETRIC, dtype=torch.float64, device=DEV)

    # --- get_actions: actions and log-probs (forward tolerance 2e-5 of the output scale)
    d_a = torch.empty(B, A, device=DEV)
    d_lp = torch.empty(B, device=DEV)
    _hip.check(L.mrl_sac_actor_sample(net, _hip.ptr(fa), _hip.ptr(d_obs), B, _hip.ptr(d_mean), _hip.ptr(d_std), -5.0, 5.0, _hip.ptr(d_ea), _hip.ptr(d_a),
                                      _hip.ptr(d_lp), _hip.ptr(ws), ws.numel(), None), 'sample')
    x = S.rms_normalize(rms, obs, hp.rms_eps)
    mu, ls = S.actor_raw(pa, x, cfg)
    a_ref, lp_ref, _ = S.squashed_sample(mu, ls, eps_a, cfg)
    assert torch.allclose(d_a.cpu(), a_ref, atol=2e-5), float((d_a.cpu() - a_ref).abs().max())
    assert torch.allclose(d_lp.cpu(), lp_ref[:, 0], rtol=2e-5, atol=2e-4), float((d_lp.cpu() - lp_ref[:, 0]).abs().max())

    # --- update_critic: loss, target statistics, gradients (3e-4 of each tensor's max, like the PPO gradients)
    gc = torch.zeros(fc.numel(), device=DEV)
    batch = _hip.Transitions(d_obs.data_ptr(), d_act.data_ptr(), d_rew.data_ptr(), d_nxt.data_ptr(), d_done.data_ptr(), O, A)
    _hip.check(L.mrl_sac_critic_grad(net, _hip.ptr(fa), _hip.ptr(fc), _hip.ptr(ft), _hip.ptr(d_la), _hip.ptr(d_mean), _hip.ptr(d_std), -5.0, 5.0, batch, B,
                                     _hip.ptr(d_en), hp.gamma ** hp.nstep, 1, _hip.ptr(gc), _hip.ptr(metrics), _hip.ptr(ws), ws.numel(), None), 'critic_grad')
    with torch.no_grad():
        nx = S.rms_normalize(rms, nxt, hp.rms_eps)
        mu2, ls2 = S.actor_raw(pa, nx, cfg)
        a2, lp2, _ = S.squashed_sample(mu2, ls2, eps_n, cfg)
        tq = S.q_min(pt, nx, a2, cfg) - log_alpha.exp() * lp2
        target = rew + (1 - done) * (hp.gamma ** hp.nstep) * tq
    leaves = S._leaves(pc)
    qs = S.q_values(leaves, x, act, cfg)
    loss = torch.sum(torch.stack([torch.nn.functional.mse_loss(q, target) for q in qs]))
    ref = dict(zip(leaves.keys(), torch.autograd.grad(loss, list(leaves.values()))))
    m = metrics.cpu()
    assert abs(float(m[0]) - float(loss)) < 2e-4 * max(1.0, abs(float(loss)))
    assert abs(float(m[1]) - float(target.mean())) < 2e-4 and abs(float(m[3]) - float(target.max())) < 2e-4 and abs(float(m[4]) - float(target.min())) < 2e-4
    got = unpack_grads(gc, offs_c, {k: tuple(v.shape) for k, v in pc.items()})
    for name, r in ref.items():
        assert rel_err(got[name], r) < 3e-4, (name, rel_err(got[name], r))

    # --- update_actor: actor loss, alpha loss, entropy, gradients of the actor and of log_alpha
    ga = torch.zeros(fa.numel(), device=DEV)
    galpha = torch.zeros(1, device=DEV)
    target_entropy = -float(A)
    _hip.check(L.mrl_sac_actor_grad(net, _hip.ptr(fa), _hip.ptr(fc), _hip.ptr(d_la), _hip.ptr(d_mean), _hip.ptr(d_std), -5.0, 5.0, _hip.ptr(d_obs), B,
                                    _hip.ptr(d_ea), target_entropy, _hip.ptr(ga), _hip.ptr(galpha), _hip.ptr(metrics), _hip.ptr(ws), ws.numel(), None), 'actor_grad')
    la = S._leaves(pa)
    mu, ls = S.actor_raw(la, x, cfg)
    a, lp, _ = S.squashed_sample(mu, ls, eps_a, cfg)
    q = S.q_min(pc, x, a, cfg)
    aloss = (log_alpha.exp() * lp - q).mean()
    ref = dict(zip(la.keys(), torch.autograd.grad(aloss, list(la.values()))))
    ref = with_head(ref)
    shapes = {k: tuple(v.shape) for k, v in with_head(pa).items()}
    got = unpack_grads(ga, offs_a, shapes)
    m = metrics.cpu()
    assert abs(float(m[0]) - float(aloss)) < 3e-4 * max(1.0, abs(float(aloss)))
    ent = -lp.detach()
    assert abs(float(m[2]) - float(ent.mean())) < 3e-4 * max(1.0, abs(float(ent.mean())))
    ll = log_alpha.clone().requires_grad_(True)
    alpha_loss = (ll.exp() * (ent - target_entropy)).mean()
    (gl,) = torch.autograd.grad(alpha_loss, [ll])
    assert abs(float(m[1]) - float(alpha_loss)) < 3e-4 * max(1.0, abs(float(alpha_loss)))
    assert abs(float(galpha.cpu()) - float(gl)) < 3e-4 * max(1.0, abs(float(gl)))
    for name in offs_a:
        assert rel_err(got[name], ref[name]) < 3e-4, (name, rel_err(got[name], ref[name]))

    # --- soft_update: one rounding per op, bit-exact
    ft2 = ft.clone()
    _hip.check(L.mrl_polyak(_hip.ptr(ft2), _hip.ptr(fc), ft2.numel(), 0.05, None), 'polyak')
    expect = ft.cpu().clone().mul_(1.0 - 0.05).add_(fc.cpu() * 0.05)
    assert torch.equal(ft2.cpu(), expect)
