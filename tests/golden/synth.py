"""Bit-reproducible synthetic inputs shared by the golden-vector generator and the tests.

Everything is derived from a 32-bit multiplicative hash of the element index, so the same
tensors are rebuilt exactly on any machine and only OUTPUTS need to be stored as fixtures.
"""

import numpy as np
import torch


def lcg(shape, seed, scale=1.0, dtype=np.float32):
    """Uniform values in [-scale/2, scale/2) from an integer hash of (seed, index); exact in float64."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    x = (idx + np.uint64(seed) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    x = (x * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    u = x.astype(np.float64) / 4294967296.0 - 0.5
    return (u * scale).astype(dtype).reshape(shape)


def lcg_t(shape, seed, scale=1.0, dtype=torch.float32):
    return torch.from_numpy(lcg(tuple(shape), seed, scale, np.float64)).to(dtype)


def gaussianish(shape, seed, dtype=torch.float32):
    """Sum of 4 uniforms, rescaled to unit variance: a smooth bell-shaped stand-in for N(0,1)."""
    s = sum(lcg(tuple(shape), seed * 4 + j, 1.0, np.float64) for j in range(4))
    return torch.from_numpy(s * np.sqrt(3.0)).to(dtype)


def make_params(names_shapes, seed):
    """Weights ~ U(-1,1)/sqrt(fan_in), LayerNorm gains near 1, small biases, in the given order."""
    out = {}
    for j, (name, shape) in enumerate(names_shapes):
        s = seed * 1000 + j
        if len(shape) == 2:
            out[name] = lcg_t(shape, s, 2.0 / np.sqrt(shape[1]))
        elif name.endswith('.weight'):  # LayerNorm gain
            out[name] = 1.0 + lcg_t(shape, s, 0.2)
        else:
            out[name] = lcg_t(shape, s, 0.1)
    return out


def actor_critic_shapes(obs_dim, act_dim, units, norm=True, act=True, fixed_sigma=False):
    """(name, shape) list in the order of the reference model's parameters()."""
    stride = 1 + int(norm) + int(act)
    res = []
    if fixed_sigma:
        res.append(('sigma', (act_dim,)))
    for tower in ('actor_mlp', 'value_mlp'):
        in_f = obs_dim
        for i, out_f in enumerate(units):
            b = i * stride
            res += [(f'{tower}.mlp.{b}.weight', (out_f, in_f)), (f'{tower}.mlp.{b}.bias', (out_f,))]
            if norm:
                res += [(f'{tower}.mlp.{b + 1}.weight', (out_f,)), (f'{tower}.mlp.{b + 1}.bias', (out_f,))]
            in_f = out_f
    res += [('value.weight', (1, units[-1])), ('value.bias', (1,))]
    res += [('mu.weight', (act_dim, units[-1])), ('mu.bias', (act_dim,))]
    if not fixed_sigma:
        res += [('sigma.weight', (act_dim, units[-1])), ('sigma.bias', (act_dim,))]
    return res


class Space:
    def __init__(self, shape):
        self.shape = tuple(shape)


class GoldenEnv:
    """Deterministic vector env: obs/reward are hash-derived and lightly coupled to the clamped actions."""

    def __init__(self, num_envs, obs_dim, act_dim, episode_length=5, seed=7, device='cpu'):
        self.num_envs, self.obs_dim, self.act_dim = num_envs, obs_dim, act_dim
        self.episode_length, self.seed, self.device = episode_length, seed, device
        self.observation_space = Space((obs_dim,))
        self.action_space = Space((act_dim,))
        self.t = 0
        self.counters = torch.arange(num_envs) % episode_length

    def _obs(self):
        return gaussianish((self.num_envs, self.obs_dim), self.seed * 100003 + self.t).to(self.device)

    def reset(self):
        self.t = 0
        self.counters = torch.arange(self.num_envs) % self.episode_length
        return self._obs()

    def step(self, actions):
        self.t += 1
        a = actions.detach().to('cpu')
        obs = self._obs().to('cpu')
        obs[:, : self.act_dim] += 0.5 * a
        rew = gaussianish((self.num_envs,), self.seed * 7919 + self.t) + 0.01 * a.sum(1)
        self.counters = self.counters + 1
        done = self.counters >= self.episode_length
        self.counters = torch.where(done, torch.zeros_like(self.counters), self.counters)
        return obs.to(self.device), rew.to(self.device), done.to(self.device), {'time_outs': done.to(self.device)}


# ---- replay-buffer scenarios (shared by make_golden_replay.py and the tests) ------------------------------------
REPLAY_NSTEP_CASES = {
    # name: (num_envs, obs_dim, act_dim, nstep, gamma, horizon of each consecutive add_to_buffer call, done probability)
    'n3': (6, 8, 4, 3, 0.99, (4, 1, 5, 2), 0.25),
    'n5': (5, 4, 2, 5, 0.97, (6, 3, 7), 0.3),
    'n1': (4, 4, 2, 1, 0.99, (3, 2), 0.25),
    'n3_nodone': (3, 4, 2, 3, 0.99, (6,), 0.0),
}
REPLAY_RING_CASES = {
    # name: (capacity, obs_dim, act_dim, rows of each consecutive add)
    'wrap': (40, 8, 4, (18, 18, 18, 7, 40)),
    'exact': (36, 4, 2, (18, 18, 5)),
    'short': (64, 4, 2, (10, 3)),
}


def replay_call_inputs(case, call, seed):
    """Trajectory of one explore call: obs / next_obs (N,T,O), actions (N,T,A), rewards (N,T,1), dones (N,T,1) float."""
    N, O, A, nstep, gamma, horizons, pdone = REPLAY_NSTEP_CASES[case]
    T = horizons[call]
    s = seed * 100 + call * 10
    obs = gaussianish((N, T, O), s + 1)
    nxt = gaussianish((N, T, O), s + 2)
    act = lcg_t((N, T, A), s + 3, 2.0)
    rew = gaussianish((N, T, 1), s + 4)
    done = (lcg_t((N, T, 1), s + 5, 1.0) + 0.5 < pdone).float()
    return obs, act, rew, nxt, done


def replay_rows(case, add, seed):
    """Flat transitions of one ReplayBuffer.add_to_buffer call: obs, actions, rewards, next_obs, dones."""
    cap, O, A, rows = REPLAY_RING_CASES[case]
    n = rows[add]
    s = seed * 100 + add * 10
    return (gaussianish((n, O), s + 1), lcg_t((n, A), s + 2, 2.0), gaussianish((n, 1), s + 3), gaussianish((n, O), s + 4),
            (lcg_t((n, 1), s + 5, 1.0) > 0.2).float())


class SacGoldenEnv(GoldenEnv):
    """GoldenEnv whose even-numbered envs end by time-out and odd-numbered ones by a real termination, so that
    `handle_timeout` (mineral/agents/agent.py:158-167) leaves some done flags set."""

    def step(self, actions):
        obs, rew, done, info = super().step(actions)
        even = (torch.arange(self.num_envs) % 2 == 0).to(done.device)
        return obs, rew, done, {'time_outs': done & even}


SAC_CASES = {
    # name: (num_envs, obs_dim, act_dim, actor units, critic units, hparams, epochs)
    'small': (8, 8, 4, [64, 32], [64, 32], dict(warm_up=4, horizon_len=4, batch_size=16, memory_size=72, nstep=3, mini_epochs=2), 3),
    'mid': (16, 12, 4, [128, 64, 32], [128, 64, 32], dict(warm_up=6, horizon_len=5, batch_size=64, memory_size=4096, nstep=3, mini_epochs=3), 2),
}


def sac_uniform(shape, counter):
    """Stand-in for torch.rand: U[0,1)."""
    return lcg_t(tuple(shape), 7000 + counter, 1.0) + 0.5


def sac_noise(shape, counter):
    """Stand-in for Normal.rsample's standard-normal draw."""
    return gaussianish(tuple(shape), 8000 + counter)


def sac_indices(high, n, counter):
    """Stand-in for torch.randint(high, (n,))."""
    u = torch.from_numpy(lcg((n,), 9000 + counter, 1.0, np.float64) + 0.5)
    return (u * high).long().clamp_(0, high - 1)


SHAC_CASES = {
    # name: (num_envs, horizon, obs_dim, critic units, n_critics, num_critic_batches, critic_iterations, optimizer, lr, max_grad_norm)
    'shac2': (16, 8, 12, [64, 64], 2, 4, 3, 'AdamW', 5e-4, 0.5),
    'single': (12, 6, 8, [32, 32], 1, 5, 2, 'Adam', 2e-3, 1.0),     # 72 samples in 5 batches: the last one is short
}


# the single `Critic` of SHAC v1 (LayerNorm + ELU towers, cfgs/agent/DFlexAntSHAC.yaml); same tuple layout, n_critics = 1
SHAC_V1_CASES = {
    'ant_v1': (16, 8, 12, [64, 64], 1, 4, 3, 'Adam', 2e-3, 1.0),
    'wide_v1': (12, 6, 8, [256, 128], 1, 5, 2, 'AdamW', 5e-4, 0.5),   # a layer of each backward kind (streamed / fused); short last batch
}


def shac_inputs(case, seed=17):
    """Rollout buffers of one SHAC epoch: obs (T,N,O), rewards, done mask, next values (T,N)."""
    N, T, O = {**SHAC_CASES, **SHAC_V1_CASES}[case][:3]
    obs = gaussianish((T, N, O), seed + 1)
    rew = gaussianish((T, N), seed + 2)
    done = (lcg_t((T, N), seed + 3, 1.0) + 0.5 < 0.15).float()
    nv = gaussianish((T, N), seed + 4) * 2.0
    return obs, rew, done, nv


# ---- Metrics with info keys (shared by make_golden_metrics.py and the tests) ------------------------------------------
# name: (num_envs, tracker_len, steps, probability that an env finishes on a step)
METRICS_CASES = {'small': (37, 10, 40, 0.15), 'wide': (3000, 100, 25, 0.1)}
METRICS_INFO_KEYS = {'sum': 'succ', 'min': 'dist', 'max': 'dist|succ', 'final': 'dist'}


def metrics_step(case, t):
    """Observation dict (`obs` plus two per-env scalars), rewards and done flags of step t."""
    N = METRICS_CASES[case][0]
    p = METRICS_CASES[case][3]
    seed = 5000 + 97 * t + (0 if case == 'small' else 13)
    obs = {'obs': gaussianish((N, 8), seed), 'succ': (lcg_t((N,), seed + 1) > 0.5).float(), 'dist': gaussianish((N,), seed + 2).abs() + 0.1}
    r = gaussianish((N,), seed + 3)
    d = lcg_t((N,), seed + 4) < (0.95 if t == 7 else p)     # one step finishes almost every env at once
    return obs, r, d
