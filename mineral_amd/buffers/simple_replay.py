"""ReplayBuffer: device-resident ring buffer with the interface of mineral/buffers/simple_replay.py:19-85."""

import torch

from .. import _hip
from . import _soa


def create_buffer(capacity, obs_space, action_dim, device='cuda'):
    """simple_replay.py:4-16 for a single flat observation key; done is uint8 (torch.bool storage)."""
    key, obs_dim = _soa.single_key(obs_space)
    rows = capacity if isinstance(capacity, int) else int(torch.tensor(capacity).prod())
    obs, action, reward, next_obs, done = _soa.alloc(rows, obs_dim, int(action_dim), device, torch.uint8)
    return {key: obs}, action, {key: next_obs}, reward, done


class ReplayBuffer:
    def __init__(self, obs_space, action_dim: int, capacity: int, device='cuda'):
        _soa.require_cuda(device)
        self.obs_space = obs_space
        self.action_dim = int(action_dim)
        self.device = device
        self.key, self.obs_dim = _soa.single_key(obs_space)
        self.next_p = 0       # next write position; like the reference it may rest at `capacity` until the next add
        self.if_full = False
        self.cur_capacity = 0
        self.capacity = int(capacity)
        self.buf_obs, self.buf_action, self.buf_next_obs, self.buf_reward, self.buf_done = create_buffer(self.capacity, obs_space, action_dim, device)
        self._ring = (self.buf_obs[self.key], self.buf_action, self.buf_reward, self.buf_next_obs[self.key], self.buf_done)
        self._cring = _soa.view(self._ring, self.obs_dim, self.action_dim)

    @torch.no_grad()
    def add_to_buffer(self, trajectory):
        """simple_replay.py:33-71.  `trajectory` = (obs dict, actions, rewards, next_obs dict, dones); any leading shape,
        flattened row-major like the reference's reshape(-1, ...)."""
        obs, actions, rewards, next_obs, dones = trajectory
        src = (obs[self.key].reshape(-1, self.obs_dim).float().contiguous(), actions.reshape(-1, self.action_dim).float().contiguous(),
               rewards.reshape(-1, 1).float().contiguous(), next_obs[self.key].reshape(-1, self.obs_dim).float().contiguous(),
               dones.reshape(-1, 1).float().contiguous())
        rows = src[2].shape[0]
        if rows > self.capacity:
            raise ValueError(f'{rows} transitions do not fit a buffer of capacity {self.capacity}')
        L = _hip.lib()
        csrc = _soa.view(src, self.obs_dim, self.action_dim)
        _hip.check(L.mrl_replay_write(self._cring, self.capacity, self.next_p, csrc, rows, _soa.stream_of(self.device)), 'mrl_replay_write')
        p = self.next_p + rows
        if p > self.capacity:
            self.if_full = True
            p -= self.capacity
        self.next_p = p
        self.cur_capacity = self.capacity if self.if_full else self.next_p

    @torch.no_grad()
    def sample_batch(self, batch_size, device='cuda', indices=None):
        """simple_replay.py:73-85.  `indices` (int64, on the device) injects the draw; by default it is torch.randint on the
        device, like the reference."""
        if indices is None:
            indices = torch.randint(self.cur_capacity, size=(batch_size,), device=self.device)
        indices = indices.to(device=self.device, dtype=torch.int64).contiguous()
        batch = int(indices.numel())
        out = _soa.alloc(batch, self.obs_dim, self.action_dim, self.device)
        L = _hip.lib()
        _hip.check(L.mrl_replay_gather(self._cring, self.capacity, indices.data_ptr(), batch, _soa.view(out, self.obs_dim, self.action_dim),
                                       _soa.stream_of(self.device)), 'mrl_replay_gather')
        obs, action, reward, next_obs, done = (t.to(device) for t in out)
        return {self.key: obs}, action, reward, {self.key: next_obs}, done
