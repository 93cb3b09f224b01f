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
    norm_kinds = (0, 2, 4, 6)      # the MRL_TOWER_* kinds with a LayerNorm (its gain / bias follow every Linear)
    if which == 0:
        names, norm = [], net.actor_kind in norm_kinds
        for l in range(net.actor_layers):
            b = (3 if norm else 2) * l
            names += [f'actor_mlp.mlp.{b}.weight', f'actor_mlp.mlp.{b}.bias'] + ([f'actor_mlp.mlp.{b + 1}.weight', f'actor_mlp.mlp.{b + 1}.bias'] if norm else [])
        return names + ['HEAD.weight', 'HEAD.bias']
    names, norm = [], net.critic_kind in norm_kinds
    for c in range(2):
        for l in range(net.critic_layers):
            b = (3 if norm else 2) * l
            names += [f'critics.{c}.mlp.{b}.weight', f'critics.{c}.mlp.{b}.bias'] + ([f'critics.{c}.mlp.{b + 1}.weight', f'critics.{c}.mlp.{b + 1}.bias'] if norm else [])
    b = (3 if norm else 2) * net.critic_layers
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


@pytest.mark.parametrize('O,A,au,cu,B', [(8, 4, [64, 32], [64, 32], 48), (64, 12, [512, 256, 128], [512, 256, 128], 300), (12, 4, [128], [256, 64], 129),
                                          (64, 12, [512, 256, 128], [512, 256, 128], 8200),   # above 8,192 rows: the split-bf16 GEMMs, incl. the input gradient through the critic
                                          (64, 12, [512, 256, 128], [512, 256, 128], 2048)])  # the batch and networks of cfgs/agent/DFlexAntSAC.yaml
def test_sac_gradients_vs_oracle(O, A, au, cu, B):
    _sac_gradients_vs_oracle(O, A, au, cu, B, ('LayerNorm', 'SiLU'), ('LayerNorm', 'SiLU'))


@pytest.mark.parametrize('O,A,au,cu,B,ak,ck', [
    (64, 12, [512, 256, 128], [512, 256, 128], 2048, ('LayerNorm', 'ELU'), ('LayerNorm', 'ELU')),   # the mlp_kwargs of DFlexAntSHAC.yaml in the SAC towers
    (64, 12, [512, 256, 128], [512, 256, 128], 300, (None, 'SiLU'), (None, 'SiLU')),
    (8, 4, [64, 32], [64, 32], 48, (None, 'ELU'), ('LayerNorm', 'ReLU')),                            # actor and critic of different kinds
    (12, 4, [128], [256, 64], 129, ('LayerNorm', 'Tanh'), (None, 'ReLU')),
    (64, 12, [256, 128], [256, 128], 8200, (None, 'SiLU'), ('LayerNorm', 'ELU'))])                   # above 8,192 rows
def test_sac_gradients_vs_oracle_other_tower_kinds(O, A, au, cu, B, ak, ck):
    """The norm_type / act_type combinations of mineral/nets/mlp.py:5-22 in the SAC actor and critics (the shipped DFlexAntSAC.yaml
    uses LayerNorm + SiLU): sampling, both losses and all gradients against the oracle's autograd."""
    _sac_gradients_vs_oracle(O, A, au, cu, B, ak, ck)


def _sac_gradients_vs_oracle(O, A, au, cu, B, ak, ck):
    from mineral_amd import _hip
    L = _hip.lib()
    cfg = S.SACConfig(O, A, au, cu, actor_kind=ak, critic_kind=ck)
    hp = S.SACHParams()
    net = _hip.make_sac_net(O, A, au, cu, actor_kind=_hip.TOWER_KINDS[ak], critic_kind=_hip.TOWER_KINDS[ck])
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
    assert abs(float(m[0]) - float(loss.detach())) < 2e-4 * max(1.0, abs(float(loss.detach())))
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
    assert abs(float(m[0]) - float(aloss.detach())) < 3e-4 * max(1.0, abs(float(aloss.detach())))
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


def _sac_cfg(case, logdir):
    from mineral_amd.config import Config
    N, O, A, au, cu, hp, epochs = synth.SAC_CASES[case]
    mlp = lambda u: dict(units=u, norm_type='LayerNorm', act_type='SiLU')
    sac = dict(multi_gpu=False, num_actors=N, reward_shaper=dict(fn='scale', scale=0.5), max_agent_steps=N * (hp['warm_up'] + epochs * hp['horizon_len']),
               tau=0.05, gamma=0.99, no_tgt_actor=True, handle_timeout=True, update_actor_interval=1, update_targets_interval=1, init_alpha=0.7,
               alpha=None, alpha_optim_kwargs=dict(lr=5e-3), backup_entropy=True, optim_type='AdamW', actor_optim_kwargs=dict(lr=5e-4),
               critic_optim_kwargs=dict(lr=5e-4), max_grad_norm=0.5)
    sac.update(hp)
    return Config.wrap(dict(
        rl_device=DEV, sim_device=DEV, multi_gpu=False, env_render=False, run='train', seed=42,
        task=dict(name='Golden', env=dict(numEnvs=N), env_autoresets=True),
        agent=dict(algo='SAC', print_every=-1, ckpt_every=-1, tracker_len=100, metrics_kwargs={},
                   network=dict(normalize_input=True, encoder=None, actor='Actor',
                                actor_kwargs=dict(mlp_kwargs=mlp(au), weight_init='orthogonal', tanh_policy=False, fixed_sigma=False,
                                                  dist_kwargs=dict(dist_type='squashed_normal', minlogstd=-5.0, maxlogstd=2.0)),
                                critic='EnsembleQ', critic_kwargs=dict(n_critics=2, mlp_kwargs=mlp(cu), weight_init='orthogonal')),
                   sac=sac)))


@pytest.mark.parametrize('case', sorted(synth.SAC_CASES))
def test_sac_agent_training_run_vs_reference(case, tmp_path):
    """The product agent (`SAC.train()`: warm-up, explore, n-step + ring buffer, update_net) against the run recorded from the
    reference agent with the same injected draws.  Replay contents are compared bit for bit where they do not depend on the
    policy (warm-up); parameters after every epoch within the tolerance of fp32 GEMMs in a different summation order."""
    from mineral_amd.agents import SAC
    G = np.load(os.path.join(ROOT, 'tests', 'golden', f'g12_sac_{case}.npz'))
    N, O, A, au, cu, hp, epochs = synth.SAC_CASES[case]
    cfg = _sac_cfg(case, str(tmp_path))
    env = synth.SacGoldenEnv(N, O, A, episode_length=5, device=DEV)
    agent = SAC(cfg, logdir=str(tmp_path), accelerator=None, datasets=None, env=env)
    agent._checkpoint_save = lambda *a, **k: None
    ocfg = S.SACConfig(O, A, au, cu)
    agent.actor.load_state_dict(synth.make_params(S.actor_shapes(ocfg), 21), strict=True)
    agent.critic.load_state_dict(synth.make_params(S.critic_shapes(ocfg), 22), strict=True)
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    counters = {'rand': 0, 'noise': 0, 'idx': 0}

    def draw(kind, fn, *a):
        counters[kind] += 1
        return fn(*a, counters[kind] - 1)

    agent.rand_fn = lambda shape: draw('rand', synth.sac_uniform, shape)
    agent.noise_fn = lambda shape: draw('noise', synth.sac_noise, shape)
    agent.index_fn = lambda high, n: draw('idx', synth.sac_indices, high, n)
    seen = []
    orig = agent.update_net

    def spy(memory):
        res = orig(memory)
        torch.cuda.synchronize()
        e = agent.epoch
        seen.append(e)
        assert [counters[k] for k in ('rand', 'noise', 'idx')] == [int(x) for x in G[f'e{e}_counters']]
        assert (memory.next_p, memory.cur_capacity, int(memory.if_full)) == tuple(int(x) for x in G[f'e{e}_memory_state'])
        k = memory.cur_capacity
        assert np.array_equal(memory.buf_done[:k].cpu().numpy(), G[f'e{e}_memory_done'])
        assert np.allclose(memory.buf_reward[:k].cpu().numpy(), G[f'e{e}_memory_reward'], rtol=0, atol=1e-5)
        if e == 0:
            # the first rows come from the warm-up (uniform actions, no policy involved): exact copies / fp32 n-step sums
            w = N * (hp['warm_up'] - hp['nstep'] + 1)
            assert np.array_equal(memory.buf_obs['obs'][:w].cpu().numpy(), G['e0_memory_obs'][:w])
            assert np.array_equal(memory.buf_action[:w].cpu().numpy(), G['e0_memory_action'][:w])
            assert np.array_equal(memory.buf_reward[:w].cpu().numpy(), G['e0_memory_reward'][:w])
            assert np.allclose(memory.buf_action[:k].cpu().numpy(), G['e0_memory_action'], atol=5e-5)
        rms = agent.obs_rms['obs']
        assert np.allclose(rms.running_mean.cpu().numpy(), G[f'e{e}_obs_rms.running_mean'], rtol=1e-5, atol=1e-6)
        assert np.allclose(rms.running_var.cpu().numpy(), G[f'e{e}_obs_rms.running_var'], rtol=1e-5, atol=1e-6)
        for name, v in agent.actor.state_dict().items():
            ref = G[f'e{e}_actor.{name}']
            assert np.allclose(v.cpu().numpy(), ref, rtol=2e-3, atol=3e-5), (e, name, float(np.abs(v.cpu().numpy() - ref).max()))
        for name, v in agent.critic.state_dict().items():
            ref = G[f'e{e}_critic.{name}']
            assert np.allclose(v.cpu().numpy(), ref, rtol=2e-3, atol=3e-5), (e, name, float(np.abs(v.cpu().numpy() - ref).max()))
        assert abs(float(agent.log_alpha.cpu()) - float(G[f'e{e}_log_alpha'])) < 2e-5
        for key in ('loss/critic', 'loss/actor', 'loss/alpha', 'entropy', 'grad_norm/critic', 'grad_norm/actor'):
            got = res[key].float().cpu().numpy()
            ref = G[f'e{e}_{key.replace("/", "_")}']
            assert np.allclose(got, ref, rtol=2e-3, atol=2e-4), (e, key, got, ref)
        if e == epochs - 1:
            for name, v in agent.critic_target.state_dict().items():
                assert np.allclose(v.cpu().numpy(), G[f'e{e}_critic_target.{name}'], rtol=2e-3, atol=3e-5), name
        return res

    agent.update_net = spy
    agent.train()
    assert seen == list(range(epochs))
    # checkpoint keys of sac.py:418-432
    ckpt = torch.load(os.path.join(str(tmp_path), 'ckpt', 'final.pth'), map_location='cpu', weights_only=True)
    assert {'epoch', 'mini_epoch', 'agent_steps', 'obs_rms', 'encoder', 'actor', 'critic', 'encoder_target', 'actor_target', 'critic_target', 'log_alpha'} <= set(ckpt)
    assert 'critics.1.mlp.0.weight' in ckpt['critic'] and 'sigma.bias' in ckpt['actor']


def test_sac_agent_at_the_baseline_configuration(tmp_path):
    """BASELINE configuration 4: SAC on the synthetic env with 4096 envs, the 1M-transition device ring, batch 2048, 8 mini-epochs,
    n-step 3 and the [512, 256, 128] networks (cfgs/agent/DFlexAntSAC.yaml through `task=Synthetic agent=SyntheticSAC`), two
    epochs after the warm-up.  With every random draw injected from a counter the run is deterministic: two runs must agree bit for
    bit (fixed-order reductions, no float atomics), all parameters stay finite, and the ring holds exactly the transitions that
    were emitted (nstep-1 per env still wait in the FIFO)."""
    from mineral_amd.agents import SAC
    from mineral_amd.envs import synthetic
    from mineral_amd.scripts import compose as C
    N = 4096

    def run():
        torch.manual_seed(123)      # the orthogonal initial weights come from torch's global generator
        tree = C.compose(os.path.join(ROOT, 'mineral_amd', 'cfgs'), ['task=Synthetic', 'agent=SyntheticSAC', f'num_envs={N}', 'agent.print_every=-1',
                                                                      'agent.ckpt_every=-1', f'logdir={tmp_path}'])
        tree['sim_device'] = tree['rl_device'] = DEV
        cfg = C.resolve(tree)
        cfg.env_seed = 1
        sac = cfg.agent.sac
        assert (int(sac.memory_size), int(sac.batch_size), int(sac.mini_epochs), int(sac.nstep)) == (1_000_000, 2048, 8, 3)
        env = synthetic.make_envs(cfg)
        agent = SAC(cfg, logdir=str(tmp_path), accelerator=None, datasets=None, env=env)
        agent._checkpoint_save = lambda *a, **k: None
        calls = [0]

        def draw(kind, shape):
            calls[0] += 1
            g = torch.Generator(device=DEV).manual_seed(1000 + calls[0])
            return torch.randn(shape, device=DEV, generator=g) if kind == 'n' else torch.rand(shape, device=DEV, generator=g)

        agent.noise_fn = lambda shape: draw('n', shape)
        agent.rand_fn = lambda shape: draw('u', shape)
        agent.index_fn = lambda high, n: (draw('u', (n,)) * high).long().clamp_(max=high - 1)
        agent.obs = agent._convert_obs(env.reset())
        agent.set_eval()
        traj, steps = agent.explore_env(env, int(sac.warm_up), random=True)
        agent.memory.add_to_buffer(traj)
        emitted = traj[1].shape[0]
        for _ in range(2):
            agent.epoch += 1
            traj, _ = agent.explore_env(env, int(sac.horizon_len), sample=True)
            agent.memory.add_to_buffer(traj)
            emitted += traj[1].shape[0]
            res = agent.update_net(agent.memory)
        torch.cuda.synchronize()
        assert emitted == N * (int(sac.warm_up) + 2 * int(sac.horizon_len) - (int(sac.nstep) - 1))
        assert agent.memory.cur_capacity == min(emitted, int(sac.memory_size))
        for k, v in res.items():
            assert torch.isfinite(torch.as_tensor(v).float()).all(), k
        return torch.cat([agent.actor.flat, agent.critic.flat, agent.critic_target.flat, agent.log_alpha]).cpu()

    a, b = run(), run()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), 'the SAC epoch is deterministic for given draws'


def test_launcher_cli_runs_sac(tmp_path):
    """`python -m mineral.scripts.run task=Synthetic agent=SyntheticSAC ...` trains SAC on the HIP path and writes final.pth."""
    import subprocess
    logdir = str(tmp_path / 'run')
    cmd = [sys.executable, '-m', 'mineral.scripts.run', 'task=Synthetic', 'agent=SyntheticSAC', 'num_envs=256', 'agent.sac.batch_size=512',
           'agent.sac.memory_size=50000', 'agent.sac.horizon_len=8', 'agent.sac.warm_up=8', 'agent.sac.mini_epochs=2',
           'agent.sac.max_agent_steps=8192', 'agent.print_every=1', f'logdir={logdir}', 'seed=5']
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'Traceback' not in res.stderr and 'Exception:' not in res.stdout, (res.stdout[-1500:], res.stderr[-1500:])
    ckpt = torch.load(os.path.join(logdir, 'ckpt', 'final.pth'), map_location='cpu', weights_only=True)
    assert ckpt['agent_steps'] == 8192 and ckpt['epoch'] == 2 and ckpt['mini_epoch'] == 5
    assert torch.isfinite(ckpt['actor']['mu.weight']).all() and torch.isfinite(ckpt['log_alpha'])
