"""Episode bookkeeping of mineral/common/metrics.py:59-67 without a host sync per env step.

The reference pulls `torch.where(dones)[0].tolist()` to the host after every env step
(mineral/agents/ppo/ppo.py:165).  Here the per-env running reward / length and the window of the last
`tracker_len` finished episodes stay on the device; the host reads them once per epoch when the means are
logged.  Window contents are identical: finished episodes enter in env order, oldest entries fall out.
"""

import torch


class DeviceTracker:
    """deque(maxlen=L) of floats on the device, initialised with zeros like tracker.py:19."""

    def __init__(self, max_len, device):
        self.max_len = max_len
        self.ring = torch.zeros(max_len + 1, dtype=torch.float64, device=device)   # slot L is a dump slot
        self.head = torch.zeros((), dtype=torch.int64, device=device)

    def push_masked(self, values, mask, rank_from_end, count):
        """Append values[mask] in index order; rank_from_end = number of masked entries after each element."""
        L = self.max_len
        keep = mask & (rank_from_end < L)
        kept = torch.clamp(count, max=L)
        # position of each kept element: head + (kept - 1 - rank_from_end)
        pos = (self.head + kept - 1 - rank_from_end) % L
        pos = torch.where(keep, pos, torch.full_like(pos, L))
        self.ring.index_put_((pos,), values.to(torch.float64))
        self.head = (self.head + kept) % L

    def mean(self):
        return float(self.ring[: self.max_len].mean().item())

    @property
    def window(self):
        L = self.max_len
        h = int(self.head.item())
        r = self.ring[:L].tolist()
        return r[h:] + r[:h]


class Metrics:
    def __init__(self, current_scores, tracker_len, info_keys=None, save_video_every=0, save_video_consecutive=0,
                 env_render=False):
        self.current_scores = current_scores
        dev = current_scores['rewards'].device
        self.tracker_len = tracker_len
        self.episode_trackers = {'rewards': DeviceTracker(tracker_len, dev), 'lengths': DeviceTracker(tracker_len, dev)}
        self._num_episodes = torch.zeros((), dtype=torch.int64, device=dev)

    @property
    def num_episodes(self):
        return int(self._num_episodes.item())

    def update(self, epoch, env, obs, rewards, dones, extras=None):
        """`dones` is the (N,) bool tensor itself (no index list, no sync)."""
        cs = self.current_scores
        cs['rewards'] += rewards
        cs['lengths'] += 1
        d = dones.to(torch.bool)
        d64 = d.to(torch.int64)
        count = d64.sum()
        rank_from_end = count - torch.cumsum(d64, 0)
        self.episode_trackers['rewards'].push_masked(cs['rewards'], d, rank_from_end, count)
        self.episode_trackers['lengths'].push_masked(cs['lengths'], d, rank_from_end, count)
        self._num_episodes += count
        cs['rewards'].masked_fill_(d, 0)
        cs['lengths'].masked_fill_(d, 0)

    def flush_video(self, epoch):
        return

    def result(self, metrics=None, prefix=''):
        return metrics if metrics is not None else {}
