"""Episode bookkeeping of mineral/common/metrics.py:59-67 without a host sync per env step.

The reference pulls `torch.where(dones)[0].tolist()` to the host after every env step
(mineral/agents/ppo/ppo.py:165).  Here the per-env running reward / length and the window of the last
`tracker_len` finished episodes stay on the device and one kernel (`mrl_episode_update`) does the whole update of an env
step; the host reads the window once per epoch when the means are logged.  Window contents are identical: finished
episodes enter in env order, oldest entries fall out.
"""

import ctypes as C

import torch

from .. import _hip


class DeviceTracker:
    """deque(maxlen=L) of floats on the device, initialised with zeros like tracker.py:19."""

    def __init__(self, max_len, device, head=None):
        self.max_len = max_len
        self.ring = torch.zeros(max_len, dtype=torch.float64, device=device)
        self.head = head if head is not None else torch.zeros(1, dtype=torch.int64, device=device)

    def mean(self):
        return float(self.ring.mean().item())

    @property
    def window(self):
        h = int(self.head.item())
        r = self.ring.tolist()
        return r[h:] + r[:h]


class Metrics:
    def __init__(self, current_scores, tracker_len, info_keys=None, save_video_every=0, save_video_consecutive=0,
                 env_render=False):
        _hip.lib()
        self.current_scores = current_scores
        dev = current_scores['rewards'].device
        if dev.type != 'cuda':
            raise _hip.HipLibraryError(f'episode metrics live on the MI355X (device={dev}); no CPU fallback')
        if current_scores['rewards'].dtype != torch.float32 or current_scores['lengths'].dtype != torch.int64:
            raise ValueError('current_scores: rewards float32, lengths int64')
        _hip.require_current_device(dev)
        self.tracker_len = tracker_len
        rewards = DeviceTracker(tracker_len, dev)
        self.episode_trackers = {'rewards': rewards, 'lengths': DeviceTracker(tracker_len, dev, head=rewards.head)}
        self._num_episodes = torch.zeros(1, dtype=torch.int64, device=dev)

    @property
    def num_episodes(self):
        return int(self._num_episodes.item())

    def update(self, epoch, env, obs, rewards, dones, extras=None):
        """`dones` is the (N,) tensor itself (no index list, no sync)."""
        cs = self.current_scores
        n = cs['rewards'].numel()
        r = rewards.reshape(-1).to(torch.float32).contiguous()
        d = dones.reshape(-1)
        if d.dtype not in (torch.bool, torch.uint8):
            d = d != 0
        d = d.contiguous()
        if r.numel() != n or d.numel() != n:
            raise ValueError(f'metrics.update: {r.numel()} rewards / {d.numel()} dones for {n} envs')
        t = self.episode_trackers
        stream = C.c_void_p(torch.cuda.current_stream(cs['rewards'].device).cuda_stream)
        _hip.check(_hip.lib().mrl_episode_update(_hip.ptr(r), _hip.ptr(d), n, _hip.ptr(cs['rewards']), _hip.ptr(cs['lengths']), _hip.ptr(t['rewards'].ring),
                                                 _hip.ptr(t['lengths'].ring), self.tracker_len, _hip.ptr(t['rewards'].head), _hip.ptr(self._num_episodes),
                                                 stream), 'mrl_episode_update')

    def flush_video(self, epoch):
        return

    def result(self, metrics=None, prefix=''):
        return metrics if metrics is not None else {}
