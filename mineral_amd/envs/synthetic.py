"""`suite: synthetic` — a random-step vectorised env that lives on the RL device.

obs ~ N(0,1) (N, obs_dim) fp32, reward ~ N(0,1) (N,), an episode ends when the per-env counter reaches
`episode_length`; counters start staggered at arange(N) % episode_length so about N / episode_length envs finish
on every step; `infos['time_outs'] = done`.  Actions are ignored.  The same class runs on CPU for the baseline.
It honours the env contract the agents consume (mineral/agents/ppo/ppo.py:151-162, agent.py:38-51).
"""

import torch


class _Space:
    def __init__(self, shape):
        self.shape = tuple(shape)


class SyntheticEnv:
    def __init__(self, num_envs, obs_dim=64, act_dim=12, episode_length=1000, device='cuda', seed=0):
        self.num_envs, self.obs_dim, self.act_dim = int(num_envs), int(obs_dim), int(act_dim)
        self.episode_length = int(episode_length)
        self.device = torch.device(device)
        self.observation_space = _Space((self.obs_dim,))
        self.action_space = _Space((self.act_dim,))
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        self.counters = torch.arange(self.num_envs, device=self.device) % self.episode_length
        # two observation buffers used in turn: the tensor a step returns stays intact until the step after the next one,
        # so an agent may still read the previous observation after calling step() (SAC stores obs and next_obs)
        self._obs2 = [torch.empty(self.num_envs, self.obs_dim, dtype=torch.float32, device=self.device) for _ in range(2)]
        self._flip = 0
        self._rew = torch.empty(self.num_envs, dtype=torch.float32, device=self.device)

    def reset(self):
        self.counters = torch.arange(self.num_envs, device=self.device) % self.episode_length
        return self._next_obs()

    def _next_obs(self):
        self._flip ^= 1
        return self._obs2[self._flip].normal_(generator=self.gen)

    def step(self, actions):
        obs = self._next_obs()
        rew = self._rew.normal_(generator=self.gen)
        self.counters += 1
        done = self.counters >= self.episode_length
        self.counters.masked_fill_(done, 0)
        return obs, rew, done, {'time_outs': done}

    def __repr__(self):
        return f'SyntheticEnv(num_envs={self.num_envs}, obs={self.obs_dim}, act={self.act_dim}, device={self.device})'


def make_envs(config):
    env_cfg = config.task.env
    return SyntheticEnv(
        num_envs=env_cfg.numEnvs,
        obs_dim=env_cfg.get('obs_dim', 64),
        act_dim=env_cfg.get('act_dim', 12),
        episode_length=env_cfg.get('episode_length', 1000),
        device=config.sim_device,
        seed=config.get('env_seed', config.seed),
    )
