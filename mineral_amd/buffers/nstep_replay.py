"""NStepReplay: n-step FIFO and masked discounted return with the interface of mineral/buffers/nstep_replay.py:10-62."""

import torch

from .. import _hip
from . import _soa


class NStepReplay:
    def __init__(self, obs_space, action_dim: int, num_envs: int = 1, nstep: int = 3, gamma: float = 0.99, device: str = 'cuda'):
        _soa.require_cuda(device)
        self.obs_space = obs_space
        self.key, self.obs_dim = _soa.single_key(obs_space)
        self.action_dim = int(action_dim)
        self.num_envs = int(num_envs)
        self.nstep = int(nstep)
        self.device = device
        if self.nstep > _hip.NSTEP_MAX:
            raise NotImplementedError(f'nstep {self.nstep} > {_hip.NSTEP_MAX}: the summation order of the reference is pinned up to {_hip.NSTEP_MAX}')
        self.nstep_count = 0
        self.gamma = gamma
        # powers in Python floats, rounded to fp32 once (nstep_replay.py:28)
        self.gamma_array = torch.tensor([self.gamma ** i for i in range(self.nstep)]).to(device).view(-1, 1)
        if self.nstep > 1:
            # the last nstep-1 transitions of every env, oldest first (zero until filled; never read before that)
            self._window = tuple(t.zero_() for t in _soa.alloc(self.num_envs * (self.nstep - 1), self.obs_dim, self.action_dim, device))
            self._cwindow = _soa.view(self._window, self.obs_dim, self.action_dim)

    @torch.no_grad()
    def add_to_buffer(self, obs, actions, rewards, next_obs, dones):
        """nstep_replay.py:31-64.  Inputs are (N, T, ...) trajectories; returns (obs dict, actions, rewards, next_obs dict,
        dones) of the emitted transitions, step-major, or the inputs unchanged when nstep == 1."""
        if self.nstep <= 1:
            return obs, actions, rewards, next_obs, dones
        N, T = dones.shape[0], dones.shape[1]
        if N != self.num_envs:
            raise ValueError(f'trajectory of {N} envs given to a buffer of {self.num_envs}')
        traj = (obs[self.key].reshape(N * T, self.obs_dim).float().contiguous(), actions.reshape(N * T, self.action_dim).float().contiguous(),
                rewards.reshape(N * T, 1).float().contiguous(), next_obs[self.key].reshape(N * T, self.obs_dim).float().contiguous(),
                dones.reshape(N * T, 1).float().contiguous())
        first = max(self.nstep - 1 - self.nstep_count, 0)
        emitted = max(T - first, 0)
        if emitted == 0:
            # the reference fails here too (torch.cat of an empty list, nstep_replay.py:59-62)
            raise ValueError('no transition emitted: fewer than nstep steps have been pushed')
        out = _soa.alloc(emitted * N, self.obs_dim, self.action_dim, self.device)
        L = _hip.lib()
        rc = L.mrl_nstep_emit(_soa.view(traj, self.obs_dim, self.action_dim), N, T, self._cwindow, self.nstep_count, self.nstep,
                              self.gamma_array.data_ptr(), _soa.view(out, self.obs_dim, self.action_dim), _soa.stream_of(self.device))
        if rc < 0:
            _hip.check(rc, 'mrl_nstep_emit')
        assert rc == emitted
        self.nstep_count += T
        o, a, r, n, d = out
        return {self.key: o}, a, r, {self.key: n}, d
