"""Episode bookkeeping of mineral/common/metrics.py:59-67 without a host sync per env step.

The reference pulls `torch.where(dones)[0].tolist()` to the host after every env step
(mineral/agents/ppo/ppo.py:165).  Here the per-env running reward / length and the window of the last
`tracker_len` finished episodes stay on the device and one kernel (`mrl_episode_update`) does the whole update of an env
step; the host reads the window once per epoch when the means are logged.  Window contents are identical: finished
episodes enter in env order, oldest entries fall out.
"""

import ctypes as C
import re

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


_INFO_MODES = {'sum': 0, 'min': 1, 'max': 2, 'final': 3}


class Metrics:
    """Episode scores and regex-selected per-env info keys (mineral/common/metrics.py:12-170) on the device.

    All windows share one ring position (the reward window's): an info value of a finished episode lands at that episode's place
    in the reward window.  That equals the reference's per-key deques as long as a key is present on every step from the
    first episode on; a key that first appears later is placed by episode, not in a deque order of its own.

    `info_keys` = {'sum' | 'min' | 'max' | 'final' | 'scalar': regex}: every key of the observation dict that matches (re.match,
    like the reference) and holds one float per env is aggregated over the episode; finished episodes enter a window of the
    last `tracker_len` values whose mean `result()` reports as `{prefix}_infos/{key}_{mode}`.  Keys whose value is not one
    number per env, and the video options, are refused loudly: they are outside the device path."""

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
        info_keys = dict(info_keys or {})
        unknown = set(info_keys) - set(_INFO_MODES) - {'scalar', 'video'}
        if unknown:
            raise ValueError(f'info_keys: unknown aggregations {sorted(unknown)}')
        if save_video_every or info_keys.get('video') is not None:
            raise NotImplementedError('video capture (save_video_every / info_keys.video, metrics.py:115-149) is outside the device path: '
                                      'render with the reference agent or leave the options unset')
        self.save_video_every, self.save_video_consecutive, self.env_render = save_video_every, save_video_consecutive, env_render
        self._info_re = {m: re.compile(info_keys.get(m, '$^')) for m in _INFO_MODES}
        self._scalar_re = re.compile(info_keys.get('scalar', '$^'))
        self.device = dev
        self.tracker_len = tracker_len
        rewards = DeviceTracker(tracker_len, dev)
        self.episode_trackers = {'rewards': rewards, 'lengths': DeviceTracker(tracker_len, dev, head=rewards.head)}
        self._num_episodes = torch.zeros(1, dtype=torch.int64, device=dev)
        self._current_info = {}    # '{key}_{mode}' -> per-env running value (metrics.py:75)
        self._info_born = {}       # '{key}_{mode}' -> episode count when the key was first seen
        self._extras = {}

    @property
    def num_episodes(self):
        return int(self._num_episodes.item())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _update_infos(self, obs, d, extras):
        """metrics.py:69-113; must run before the episode rings' head advances for this step."""
        for k, v in (extras or {}).items():
            if self._scalar_re.match(k):
                self._extras[k] = v.item() if hasattr(v, 'item') else float(v)      # metrics.py:71-73
        if not isinstance(obs, dict):
            return
        n = self.current_scores['rewards'].numel()
        for k, v in obs.items():
            for mode, code in _INFO_MODES.items():
                if not self._info_re[mode].match(k):
                    continue
                if not isinstance(v, torch.Tensor):
                    v = torch.as_tensor(v)
                if v.numel() != n:
                    raise NotImplementedError(f'info key `{k}` has shape {tuple(v.shape)}: only one value per env is aggregated on the device')
                x = v.reshape(-1).to(self.device, torch.float32).contiguous()
                name = f'{k}_{mode}'
                if name not in self._current_info:
                    init = float('nan') if mode in ('min', 'max') else 0.0        # fmin / fmax skip NaN: the first value is taken as it is
                    self._current_info[name] = torch.full((n,), init, dtype=torch.float32, device=self.device)
                    self.episode_trackers[name] = DeviceTracker(self.tracker_len, self.device, head=self.episode_trackers['rewards'].head)
                    # episodes finished before this key was first seen: the reference creates a key's Tracker when an episode first
                    # finishes with it (metrics.py:104-107), so result() leaves the key out until then
                    self._info_born[name] = self._num_episodes.clone()
                # metrics.py:110-113 picks the restart value by the NAME: 0 when it contains 'sum' or 'final', NaN otherwise
                reset_nan = 0 if ('sum' in name or 'final' in name) else 1
                _hip.check(_hip.lib().mrl_info_track(_hip.ptr(x), _hip.ptr(d), n, code, reset_nan, _hip.ptr(self._current_info[name]),
                                                     _hip.ptr(self.episode_trackers[name].ring), self.tracker_len,
                                                     _hip.ptr(self.episode_trackers['rewards'].head), self._stream()), 'mrl_info_track')

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
        self._update_infos(obs, d, extras)
        t = self.episode_trackers
        _hip.check(_hip.lib().mrl_episode_update(_hip.ptr(r), _hip.ptr(d), n, _hip.ptr(cs['rewards']), _hip.ptr(cs['lengths']), _hip.ptr(t['rewards'].ring),
                                                 _hip.ptr(t['lengths'].ring), self.tracker_len, _hip.ptr(t['rewards'].head), _hip.ptr(self._num_episodes),
                                                 self._stream()), 'mrl_episode_update')

    def flush_video(self, epoch):
        return   # video options are refused at construction

    def result(self, metrics=None, prefix=''):
        """metrics.py:151-170."""
        metrics = {} if metrics is None else metrics
        for k, v in self.episode_trackers.items():
            if k in self.current_scores:          # rewards and lengths are reported by the agents themselves
                continue
            if k in self._info_born and int(self._num_episodes.item()) <= int(self._info_born[k].item()):
                continue                          # no episode has finished since the key appeared: the reference has no Tracker for it yet
            metrics[f'{prefix}_infos/{k}'] = v.mean()
        for k, v in self._extras.items():
            metrics[f'{prefix}_extras/{k}'] = v
        self._extras.clear()
        return metrics
