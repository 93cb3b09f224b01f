"""SAC on the synthetic env at the baseline's configuration (4096 envs, 1M-transition device replay buffer, batch 2048,
8 mini-epochs, horizon 32, nstep 3, MLP [512,256,128] LayerNorm+SiLU): ms per epoch and env-steps/s of the HIP agent, with
the CPU oracle's update loop timed beside it on the host cores.

    python tools/sac_bench.py [out.json] [--epochs K]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import torch  # noqa: E402

N, T, B, MINI, CAP = 4096, 32, 2048, 8, 1_000_000
O, A, UNITS = 64, 12, [512, 256, 128]


def make_cfg(dev):
    from mineral_amd.scripts import compose as C
    tree = C.compose(os.path.join(ROOT, 'mineral_amd', 'cfgs'), ['task=Synthetic', 'agent=SyntheticSAC', f'num_envs={N}', 'agent.print_every=-1',
                                                                  'agent.ckpt_every=-1', 'logdir=/tmp/mrl_sac_bench'])
    tree['sim_device'] = tree['rl_device'] = dev
    return C.resolve(tree)


def cpu_baseline(threads):
    """One update_net of the oracle (8 mini-epochs at batch 2048) plus one policy step for 4096 envs, on `threads` host threads."""
    import synth
    from oracle import ppo_oracle as P
    from oracle import sac_oracle as S
    torch.set_num_threads(threads)
    cfg = S.SACConfig(O, A, UNITS, UNITS)
    hp = S.SACHParams()
    pa, pc = synth.make_params(S.actor_shapes(cfg), 1), synth.make_params(S.critic_shapes(cfg), 2)
    pt = {k: v.clone() for k, v in pc.items()}
    la = torch.tensor(0.0)
    opts = {'actor': P.adam_state_init(pa), 'critic': P.adam_state_init(pc), 'alpha': P.adam_state_init({'log_alpha': la})}
    rms = S.rms_init((O,))
    g = torch.Generator().manual_seed(0)
    batch = (torch.randn(B, O, generator=g), torch.rand(B, A, generator=g) * 2 - 1, torch.randn(B, 1, generator=g), torch.randn(B, O, generator=g),
             (torch.rand(B, 1, generator=g) < 0.01).float())
    obs = torch.randn(N, O, generator=g)
    t0 = time.perf_counter()
    S.update_net(pa, pc, pt, la, opts, rms, lambda i: batch, lambda i, w: torch.randn(B, A, generator=g), cfg, hp)
    t_upd = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(4):
        S.get_actions(pa, rms, obs, torch.randn(N, A, generator=g), cfg, hp)
    t_act = (time.perf_counter() - t0) / 4
    epoch = t_upd + T * t_act
    return {'value': N * T / epoch, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'sample': f'1 update_net ({MINI} mini-epochs x batch {B}) {t_upd:.2f} s + {T} policy steps of {N} envs ({t_act * 1e3:.0f} ms each, 4 timed); '
                      'env stepping and replay excluded'}


def main():
    out_path = next((a for a in sys.argv[1:] if not a.startswith('--')), None)
    epochs = int(sys.argv[sys.argv.index('--epochs') + 1]) if '--epochs' in sys.argv else 10
    dev = 'cuda:0'
    from mineral_amd.agents import SAC
    from mineral_amd.envs import synthetic
    cfg = make_cfg(dev)
    cfg.env_seed = 1
    env = synthetic.make_envs(cfg)
    agent = SAC(cfg, logdir='/tmp/mrl_sac_bench', accelerator=None, datasets=None, env=env)
    agent._checkpoint_save = lambda *a, **k: None
    agent.obs = agent._convert_obs(env.reset())
    agent.set_eval()
    traj, steps = agent.explore_env(env, cfg.agent.sac.warm_up, random=True)
    agent.memory.add_to_buffer(traj)

    def one_epoch():
        agent.epoch += 1
        traj, _ = agent.explore_env(env, T, sample=True)
        agent.memory.add_to_buffer(traj)
        return agent.update_net(agent.memory)

    for _ in range(3):
        one_epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        one_epoch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / epochs
    # the update alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        agent.update_net(agent.memory)
    torch.cuda.synchronize()
    du = (time.perf_counter() - t0) / epochs
    macs_fwd = 2 * 0 + (O * 512 + 512 * 256 + 256 * 128 + 128 * 2 * A)            # actor
    macs_q = 2 * ((O + A) * 512 + 512 * 256 + 256 * 128 + 128)                     # both critics
    # per mini-epoch and sample: actor fwd x2 + bwd, critics fwd x3 + bwd (params) + bwd (inputs)
    flop_update = MINI * B * 2.0 * (2 * macs_fwd + 2 * macs_fwd + 3 * macs_q + 2 * macs_q + 2 * macs_q)
    res = {
        'metric': 'SAC env-steps/sec, synthetic env (obs=64, act=12), 4096 envs, 1M-transition device replay buffer',
        'value': N * T / dt, 'unit': 'env-steps/s', 'ms_per_epoch': dt * 1e3, 'ms_update_net': du * 1e3,
        'config': {'workload': f'SAC DFlexAntSAC hparams: {N} envs x horizon {T}, batch {B}, {MINI} mini-epochs, nstep 3, ring of {CAP}'},
        'roofline': {'bound': 'launch latency (about 40 small launches per mini-epoch at batch 2048)', 'update_TFLOPs': flop_update / du / 1e12,
                     'peak': 157.3, 'frac': flop_update / du / 1e12 / 157.3},
        'cpu_baseline': cpu_baseline(min(16, os.cpu_count() or 1)),
    }
    print(json.dumps(res, indent=1))
    if out_path:
        json.dump(res, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
