"""Drives the SAC oracle through the same run that tests/golden/make_golden_sac.py recorded from the reference agent
(`SAC.train()`, mineral/agents/sac/sac.py:190-249): warm-up with uniform actions, then epochs of explore + update_net,
with every random draw taken from synth.py in the order the reference consumed them."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import synth  # noqa: E402
from oracle import ppo_oracle as P  # noqa: E402
from oracle import replay_oracle as R  # noqa: E402
from oracle import sac_oracle as S  # noqa: E402


def case_setup(case):
    N, O, A, au, cu, hp, epochs = synth.SAC_CASES[case]
    cfg = S.SACConfig(O, A, au, cu)
    h = S.SACHParams(batch_size=hp['batch_size'], tau=0.05, nstep=hp['nstep'], mini_epochs=hp['mini_epochs'], init_alpha=0.7)
    return N, O, A, cfg, h, hp, epochs


def run_oracle(case, on_epoch=None):
    N, O, A, cfg, h, hp, epochs = case_setup(case)
    env = synth.SacGoldenEnv(N, O, A, episode_length=5)
    pa = synth.make_params(S.actor_shapes(cfg), 21)
    pc = synth.make_params(S.critic_shapes(cfg), 22)
    pt = {k: v.clone() for k, v in pc.items()}
    log_alpha = torch.tensor(np.log(h.init_alpha), dtype=torch.float32)
    opts = {'actor': P.adam_state_init(pa), 'critic': P.adam_state_init(pc), 'alpha': P.adam_state_init({'log_alpha': log_alpha})}
    obs_rms = S.rms_init((O,), h.rms_eps)
    nbuf = R.NStepOracle(N, O, A, h.nstep, h.gamma)
    mem = R.RingOracle(O, A, hp['memory_size'])
    counters = {'rand': 0, 'noise': 0, 'idx': 0}

    def draw(kind, *a):
        counters[kind] += 1
        c = counters[kind] - 1
        return {'rand': synth.sac_uniform, 'noise': synth.sac_noise, 'idx': lambda hi, n, c_: synth.sac_indices(hi, n, c_)}[kind](*a, c)

    state = {'obs': env.reset(), 'obs_rms': obs_rms}

    def explore(timesteps, random):
        t_obs, t_nxt = torch.empty(N, timesteps, O), torch.empty(N, timesteps, O)
        t_act, t_rew, t_done = torch.empty(N, timesteps, A), torch.empty(N, timesteps), torch.empty(N, timesteps)
        for i in range(timesteps):
            state['obs_rms'] = P.rms_update(state['obs_rms'], state['obs'])                  # sac.py:155-157
            if random:
                actions = draw('rand', (N, A)) * 2.0 - 1.0                                   # sac.py:158-159
            else:
                actions = S.get_actions(pa, state['obs_rms'], state['obs'], draw('noise', (N, A)), cfg, h)
            nxt, rew, dones, infos = env.step(actions)
            dones = dones * (~infos['time_outs'])                                            # handle_timeout, agent.py:158-167
            t_obs[:, i], t_act[:, i], t_done[:, i], t_rew[:, i], t_nxt[:, i] = state['obs'], actions, dones, rew, nxt
            state['obs'] = nxt
        t_rew = (t_rew * 0.5).reshape(N, timesteps, 1)                                       # reward_shaper scale 0.5
        out = nbuf.add_to_buffer(t_obs.numpy(), t_act.numpy(), t_rew.numpy(), t_nxt.numpy(), t_done.reshape(N, timesteps, 1).numpy())
        mem.add_to_buffer(*out)

    explore(hp['warm_up'], True)
    mini_epoch = -1
    for e in range(epochs):
        explore(hp['horizon_len'], False)

        def sample_fn(i):
            idx = draw('idx', mem.cur_capacity, h.batch_size).numpy()
            return tuple(torch.from_numpy(x) for x in mem.gather(idx))

        def noise_fn(i, which):
            return draw('noise', (h.batch_size, A))

        rec = []
        mini_epoch = S.update_net(pa, pc, pt, log_alpha, opts, state['obs_rms'], sample_fn, noise_fn, cfg, h, mini_epoch, rec)
        if on_epoch is not None:
            on_epoch(e, dict(pa=pa, pc=pc, pt=pt, log_alpha=log_alpha, obs_rms=state['obs_rms'], mem=mem, rec=rec, counters=dict(counters)))
    return dict(pa=pa, pc=pc, pt=pt, log_alpha=log_alpha, obs_rms=state['obs_rms'], mem=mem)
