"""CPU-side checks: the C ABI loads and exports what include/mineral_ppo.h declares, argument validation works
without a GPU, the host mirrors of the reference interfaces behave, and the sharded (N>1) path is correct under
gloo with world_size 2."""

import collections
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth
from oracle import ppo_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    names = set()
    for header in ('mineral_ppo.h', 'mineral_replay.h', 'mineral_sac.h', 'mineral_shac.h'):
        text = open(os.path.join(ROOT, 'include', header)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        names |= set(re.findall(r'\b(mrl_[a-z0-9_]+)\s*\(', text))
    return sorted(names)


def test_abi_exports_every_declared_symbol():
    from mineral_amd import _hip
    lib = _hip.lib()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} is declared in include/*.h but not exported'
    assert sorted(_hip.EXPORTS) == names, 'the ctypes binding and the header must list the same entry points'
    assert lib.mrl_version() >= 1


def test_abi_argument_validation_without_gpu():
    from mineral_amd import _hip
    L = _hip.lib()
    net = _hip.make_net(64, 12, [512, 256, 128], 'dreamerv3_normal', 0.1, 1.0)
    assert L.mrl_param_count(net) == 401820        # 401,817 parameters + 3 floats of alignment padding
    assert L.mrl_workspace_bytes(net, 32768) > 0
    bad = _hip.make_net(63, 12, [512, 256, 128], 'dreamerv3_normal')
    assert L.mrl_param_count(bad) == -1
    assert b'obs_dim' in L.mrl_last_error()
    bad = _hip.make_net(64, 12, [500], 'normal')
    assert L.mrl_workspace_bytes(bad, 10) == -1
    with pytest.raises(ValueError):
        _hip.check(L.mrl_rms_merge(None, None, None, 1.0, None, None, None), 'mrl_rms_merge')
    with pytest.raises(ValueError):
        _hip.check(L.mrl_adam_step(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 1, 0.0, 1.0, None, None, None), 'adam')
    with pytest.raises(NotImplementedError):
        _hip.make_net(64, 12, [32], 'squashed_normal')


def test_missing_library_fails_loudly(monkeypatch):
    from mineral_amd import _hip
    monkeypatch.setattr(_hip, '_lib', None)
    monkeypatch.setattr(_hip, 'LIB_PATH', '/nonexistent/libmineral_ppo.so')
    with pytest.raises(_hip.HipLibraryError):
        _hip.lib()


def test_parameter_views_match_reference_names_and_count():
    from mineral_amd.agents.ppo.models import ActorCritic
    m = ActorCritic({'obs': (64,)}, 12, mlp_kwargs=dict(units=[512, 256, 128], norm_type='LayerNorm', act_type='SiLU'),
                    fixed_sigma=False, actor_dist_kwargs=dict(dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0), device='cpu')
    shapes = synth.actor_critic_shapes(64, 12, [512, 256, 128], True, True, False)
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in shapes]
    assert all(tuple(v.shape) == s for (_, v), (_, s) in zip(m.named_parameters(), shapes))
    assert sum(v.numel() for v in m.parameters()) == 401817
    # init statistics of models.py:75-91: orthogonal rows, zero biases, unit LayerNorm gains
    sd = m.state_dict()
    w = sd['actor_mlp.mlp.3.weight']
    assert torch.allclose(w @ w.t(), 2.0 * torch.eye(256), atol=1e-4)
    assert torch.allclose(sd['mu.weight'] @ sd['mu.weight'].t(), 1e-4 * torch.eye(12), atol=1e-7)
    assert float(sd['value.bias'].abs().sum()) == 0 and torch.all(sd['value_mlp.mlp.1.weight'] == 1)
    # views alias the flat vector
    params = synth.make_params(shapes, 3)
    m.load_state_dict(params)
    flat2 = m.flat.clone()
    assert torch.equal(m.views_of(flat2)['sigma.bias'], params['sigma.bias'])
    # the network of CartpolePPO.yaml: no normalisation layer (Linear at every second index), ELU, `sigma` a direct parameter
    c = ActorCritic({'obs': (4,)}, 1, mlp_kwargs=dict(units=[32, 32], norm_type=None, act_type='ELU'), fixed_sigma=True,
                    actor_dist_kwargs=dict(dist_type='normal'), device='cpu')
    cshapes = synth.actor_critic_shapes(4, 1, [32, 32], False, True, True)
    assert [n for n, _ in c.named_parameters()] == [n for n, _ in cshapes]
    assert all(tuple(v.shape) == s for (_, v), (_, s) in zip(c.named_parameters(), cshapes))
    assert float(c.state_dict()['sigma'].abs().sum()) == 0            # models.py:87-88
    e = ActorCritic({'obs': (64,)}, 12, mlp_kwargs=dict(units=[32], norm_type='LayerNorm', act_type='ELU'), fixed_sigma=True, device='cpu')
    assert e.net.tower_kind == 2 and 'mu.weight' in e.state_dict()    # LayerNorm + ELU towers: same layout as LayerNorm + SiLU
    # every norm_type in {None, LayerNorm} x act_type in {SiLU, ELU, ReLU, Tanh} of mlp.py:5-22 has kernels
    t = ActorCritic({'obs': (64,)}, 12, mlp_kwargs=dict(units=[32], norm_type='LayerNorm', act_type='Tanh'), fixed_sigma=True, device='cpu')
    r = ActorCritic({'obs': (64,)}, 12, mlp_kwargs=dict(units=[32], norm_type=None, act_type='ReLU'), fixed_sigma=False, device='cpu')
    assert (t.net.tower_kind, r.net.tower_kind) == (6, 5) and 'actor_mlp.mlp.1.weight' in t.state_dict() and 'actor_mlp.mlp.1.weight' not in r.state_dict()
    with pytest.raises(NotImplementedError):                           # pairs without kernels
        ActorCritic({'obs': (64,)}, 12, mlp_kwargs=dict(units=[32], norm_type='BatchNorm1d', act_type='ReLU'), fixed_sigma=True, device='cpu')
    with pytest.raises(NotImplementedError):
        ActorCritic({'obs': (64,)}, 12, mlp_kwargs=dict(units=[32], norm_type=None, act_type='GELU'), fixed_sigma=False, device='cpu')


def test_schedulers_match_reference_tables(golden):
    from mineral_amd.agents.ppo.utils import AdaptiveScheduler, LinearScheduler, adjust_learning_rate_cos
    g = golden('g9_g10_seeds_sched')
    sch = AdaptiveScheduler(0.008, 1e-6, 1e-2)
    lr, lrs = 5e-4, []
    for k in synth.lcg((40,), 77, 0.04, np.float64) + 0.02:
        lr = sch.update(lr, float(k))
        lrs.append(lr)
    assert np.array_equal(np.array(lrs), g['kl_lrs'])
    lin = LinearScheduler(5e-4, 1e-6, 1000000)
    assert np.array_equal(np.array([lin.update(s) for s in range(0, 1200000, 100000)]), g['lin_lrs'])
    cos = [adjust_learning_rate_cos(5e-4, e, 5, s, 4100000) for e in range(5) for s in (0, 1000000, 4000000)]
    assert np.array_equal(np.array(cos), g['cos_lrs'])


def test_config_tree_access_patterns():
    from mineral_amd.config import Config
    c = Config.wrap({'agent': {'ppo': {'e_clip': 0.2, 'optim_kwargs': {'lr': 5e-4}}}, 'x': [1, {'y': 2}]})
    assert c.agent.ppo.e_clip == 0.2 and c['agent']['ppo'].get('missing', 7) == 7 and c.agent.ppo.optim_kwargs.get('lr') == 5e-4
    assert c.x[1].y == 2
    with pytest.raises(AttributeError):
        c.nope
    assert c.to_container()['agent']['ppo']['optim_kwargs'] == {'lr': 5e-4}


def test_timer_statistics():
    from mineral_amd.common.timer import Timer
    t = Timer()

    class Obj:
        def work(self):
            return 3

    o = Obj()
    t.wrap('agent', o, ['work'])
    assert o.work() == 3 and o.work() == 3
    st = t.stats(step=100, total_names=('agent.work',), reset=False)
    assert st['agent.work/count'] == 2 and st['totalrate'] > 0 and 'agent.work/avg' in st


# ---------------------------------------------------------------------------------------------------------------------
# world_size 2 under gloo: moment merging and gradient averaging of the sharded path
# ---------------------------------------------------------------------------------------------------------------------
def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from mineral_amd import parallel
    try:
        # (1) per-minibatch observation moments: each rank holds its own rows of every minibatch
        n_mb, mb, O_ = 3, 40, 6
        full = synth.gaussianish((n_mb, world * mb, O_), 31) * 2.0 + 0.7
        mine = full[:, rank * mb:(rank + 1) * mb]
        mean, var = mine.mean(1).float(), mine.var(1, correction=1).float()
        cnt = parallel.combine_moments(mean, var, mb, dist, world)
        assert cnt == world * mb
        np.testing.assert_allclose(mean.numpy(), full.mean(1).numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(var.numpy(), full.var(1, correction=1).numpy(), rtol=1e-5, atol=1e-6)
        # every rank must hold the same bits, or the replicas' normalisers drift apart
        both = [torch.empty_like(var) for _ in range(world)]
        dist.all_gather(both, var)
        assert torch.equal(both[0], both[1])

        # (1b) the fused per-update exchange: GAE sums and minibatch moments in one all_gather, KL in the gradient's tail
        sums = torch.arange(6, dtype=torch.float64) * (rank + 1) + 0.25
        mean2, var2 = mine.mean(1).float(), mine.var(1, correction=1).float()
        cnt = parallel.exchange_update_statistics(sums, mean2, var2, mb, dist, world)
        assert cnt == world * mb and torch.equal(sums, torch.arange(6, dtype=torch.float64) * 3 + 0.5)
        assert torch.equal(mean2, mean) and torch.equal(var2, var), 'same merge as combine_moments, bit for bit'
        only = torch.ones(6, dtype=torch.float64) * (rank + 2)
        assert parallel.exchange_update_statistics(only, None, None, mb, dist, world) == mb and float(only[0]) == 5.0
        grads = torch.full((10 + parallel.GRAD_TAIL,), float(rank + 1))
        kl = torch.tensor(0.1234567890123456 * (rank + 1), dtype=torch.float64)
        parallel.pack_scalar(grads[10:], kl)
        parallel.average_gradients(grads, dist)
        assert torch.equal(grads[:10], torch.full((10,), 3.0))
        assert abs(float(parallel.unpack_scalar(grads[10:])) - 0.1234567890123456 * 3) < 1e-7      # an fp32 sum over the ranks
        one = torch.zeros(parallel.GRAD_TAIL)
        parallel.pack_scalar(one, kl)
        assert abs(float(parallel.unpack_scalar(one)) - float(kl)) < 1e-14                               # exact enough to be invisible with one rank

        # (2) one optimizer step: shard-local gradients, summed over ranks and scaled by 1/W, equal the gradient of the
        # union minibatch (the DDP all-reduce of mineral/agents/ppo/ppo.py:292)
        spec = dict(obs=8, act=3, units=[16, 8])
        shapes = synth.actor_critic_shapes(spec['obs'], spec['act'], spec['units'], True, True, False)
        cfg = O.NetConfig(spec['obs'], spec['act'], spec['units'], 'LayerNorm', 'SiLU', False, 'dreamerv3_normal', 0.1, 1.0)
        hp = O.PPOHParams(minibatch_size=32, normalize_input=False)
        Bt = 64
        data = {'obs': synth.gaussianish((Bt, 8), 1), 'actions': synth.gaussianish((Bt, 3), 2) * 0.5,
                'mu': torch.tanh(synth.gaussianish((Bt, 3), 3)), 'sigma': synth.lcg_t((Bt, 3), 4) * 0.5 + 0.5,
                'neglogp': synth.gaussianish((Bt,), 5) * 0.3 + 3.0, 'advantages': synth.gaussianish((Bt,), 6),
                'returns': synth.gaussianish((Bt, 1), 7), 'values': synth.gaussianish((Bt, 1), 8)}
        shard = {k: v[rank * 32:(rank + 1) * 32] for k, v in data.items()}

        def hook(grads):
            flat = torch.cat([g.reshape(-1) for g in grads.values()])
            parallel.average_gradients(flat, dist)
            flat *= 1.0 / world
            out, o = collections.OrderedDict(), 0
            for k, g in grads.items():
                out[k] = flat[o:o + g.numel()].view_as(g)
                o += g.numel()
            return out

        p_shard = synth.make_params(shapes, 2)
        O.minibatch_step(p_shard, O.adam_state_init(p_shard), None, shard, cfg, hp, 5e-4, grad_hook=hook)
        p_full = synth.make_params(shapes, 2)
        hp_full = O.PPOHParams(minibatch_size=64, normalize_input=False)
        O.minibatch_step(p_full, O.adam_state_init(p_full), None, data, cfg, hp_full, 5e-4)
        for k in p_full:
            np.testing.assert_allclose(p_shard[k].numpy(), p_full[k].numpy(), rtol=2e-5, atol=1e-7, err_msg=k)
        q.put((rank, 'ok'))
    except Exception as e:   # surface the failure in the parent
        q.put((rank, f'{type(e).__name__}: {e}'))
    finally:
        dist.destroy_process_group()


def test_sharded_path_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, 'ok'), (1, 'ok')], results


def test_host_classes_accept_the_dflex_ant_shapes():
    """obs=37 / act=8 (DFlex Ant): the parameter container, the rollout buffer and the running statistics are built with
    observation rows padded to 40 columns for the kernels while every tensor under a reference name keeps its shape."""
    import torch
    from mineral_amd.agents.ppo.experience import ExperienceBuffer
    from mineral_amd.agents.ppo.models import ActorCritic
    m = ActorCritic({'obs': (37,)}, 8, mlp_kwargs=dict(units=[512, 256, 128], norm_type='LayerNorm', act_type='SiLU'), fixed_sigma=False,
                    actor_dist_kwargs=dict(dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0), device='cpu')
    assert (m.obs_dim, m.obs_pad, m.net.obs_dim) == (37, 40, 40)
    sd = m.state_dict()
    assert sd['actor_mlp.mlp.0.weight'].shape == (512, 37) and sd['value_mlp.mlp.0.weight'].shape == (512, 37)
    assert sum(v.numel() for v in sd.values()) == 2 * (37 * 512 + 512 * 256 + 256 * 128 + 2 * (512 + 256 + 128) + 512 + 256 + 128) + 128 * 17 + 17
    w = torch.arange(512 * 37, dtype=torch.float32).view(512, 37)
    sd['actor_mlp.mlp.0.weight'] = w
    m.load_state_dict(sd)
    assert torch.equal(m.state_dict()['actor_mlp.mlp.0.weight'], w)
    assert float(m.flat[: 512 * 40].view(512, 40)[:, 37:].abs().max()) == 0.0
    buf = ExperienceBuffer(4, 3, 12, 12, {'obs': (37,)}, 8, 'cpu')
    assert buf.storage_dict['obses']['obs'].shape == (4, 3, 37) and buf.obs_store.shape == (4, 3, 40)
    buf.storage_dict['obses']['obs'].copy_(torch.ones(4, 3, 37))
    assert float(buf.obs_store[:, :, 37:].abs().max()) == 0.0
    assert buf.prepare_training()['obses']['obs'].shape == (12, 37)


def test_custom_operators_are_registered_for_the_device_only():
    """mineral_amd/ops.py registers the C ABI's entry points under torch.ops.mineral for the CUDA (ROCm) dispatch key; the
    host-only queries answer here, a call with CPU tensors is refused by the dispatcher (no CPU fallback)."""
    import torch
    import mineral_amd.ops as ops
    for n in ops.OPS:
        assert hasattr(torch.ops.mineral, n), n
    net_i, net_f = [64, 12, 1, 0, 0, 512, 256, 128], [0.1, 1.0, 1e-5]
    assert torch.ops.mineral.param_count(net_i, net_f) == 401820
    assert torch.ops.mineral.workspace_bytes(net_i, net_f, 2048) > 0
    with pytest.raises(ValueError):
        torch.ops.mineral.param_count([63, 12, 1, 0, 0, 512], net_f)
    with pytest.raises(NotImplementedError):
        torch.ops.mineral.clip_adam_step_(torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4), 1e-3, 0.9, 0.999, 1e-8, 0.0, False, 1, 0.0, 1.0,
                                          None, torch.zeros(4096, dtype=torch.uint8))


def test_template_agent_subclass_constructs_and_base_helpers(tmp_path):
    """mineral/agents/template_agent.py: the smallest subclass goes through the base constructor (agent.py:13-81); the base
    class's helpers `_as_metrics`, `_handle_timeout`, `_convert_obs` and the abstract `get_actions` / `explore_env` behave as
    agent.py:83-167.  The episode metrics live on the GPU, so this CPU test gives the subclass a stand-in for them (the
    constructor's call of `_create_metrics` is still exercised) and checks that the real ones refuse a CPU device loudly."""
    import types
    import torch
    from mineral_amd import _hip, agents
    from mineral_amd.config import Config

    class Space:
        def __init__(self, shape):
            self.shape = shape

    env = types.SimpleNamespace(action_space=Space((3,)), observation_space=Space((5,)))
    cfg = Config.wrap(dict(rl_device='cpu', multi_gpu=False, env_render=False, task={},
                           agent=dict(algo='Template', network=dict(normalize_input=True, cpu_obs_keys='^img'), template=dict(num_actors=4, max_agent_steps=100),
                                      print_every=1)))
    assert agents.Template is getattr(agents, cfg.agent.algo)
    with pytest.raises(_hip.HipLibraryError):        # the real metrics: device only, no CPU fallback
        agents.Template(cfg, logdir=str(tmp_path / 'a'), env=env)

    class Mine(agents.Template):
        def _create_metrics(self, tracker_len, metrics_kwargs):
            return ('metrics', tracker_len)

    a = Mine(cfg, logdir=str(tmp_path / 'b'), accelerator=None, datasets=None, env=env)
    assert a.num_actors == 4 and a.max_agent_steps == 100 and a.action_dim == 3 and a.obs_space == {'obs': (5,)}
    assert a.metrics == ('metrics', 100) and a.normalize_input is True and (a.epoch, a.mini_epoch, a.agent_steps) == (-1, -1, 0)
    assert os.path.isdir(os.path.join(str(tmp_path / 'b'), 'ckpt'))
    for call in (lambda: a.get_actions(None), lambda: a.explore_env(None, 1)):
        with pytest.raises(NotImplementedError):
            call()
    assert a.train() is None and a.eval() is None and a.save('x') is None and a.load('x') is None
    with a._as_metrics('eval metrics'):
        assert a.metrics == 'eval metrics'
    assert a.metrics == ('metrics', 100)
    with pytest.raises(RuntimeError):
        with a._as_metrics('eval metrics'):
            raise RuntimeError
    assert a.metrics == ('metrics', 100)
    dones = torch.tensor([True, True, False, True])
    trunc = torch.tensor([False, True, False, False])
    assert torch.equal(agents.Agent._handle_timeout(dones, {'TimeLimit.truncated': trunc}), torch.tensor([True, False, False, True]))
    assert torch.equal(agents.Agent._handle_timeout(dones, {'time_outs': ~trunc, 'truncated': trunc}), dones * trunc)   # the first key found decides
    assert agents.Agent._handle_timeout(dones, {}) is dones
    o = a._convert_obs({'obs': np.zeros((4, 5), np.float32), 'img0': np.zeros((4, 2), np.float32), 't': torch.ones(4)})
    assert o['obs'].device.type == 'cpu' and o['img0'].device.type == 'cpu' and o['t'].shape == (4,)
    assert set(a._convert_obs(np.zeros((4, 5), np.float32))) == {'obs'}
