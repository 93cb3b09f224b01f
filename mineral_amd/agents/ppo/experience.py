"""Device-resident PPO rollout storage (interface of mineral/agents/ppo/experience.py).

Per-sample vectors (obs, actions, mu, sigma) are stored ENV-MAJOR, `[N][T][.]`, so the flattened training
view `(N*T, .)` with flat index n*T + t — what the reference builds with transpose(0,1)+reshape
(experience.py:7-14) — is the storage itself and minibatch i is the contiguous row range
[i*mb, (i+1)*mb) (experience.py:61-64).  Per-step scalars are staged time-major `[T][N]` while collecting
and are flattened by the GAE kernel.
"""

import torch

from ... import _hip


class ExperienceBuffer:
    def __init__(self, num_envs, horizon_len, batch_size, minibatch_size, obs_space, act_dim, device, cpu_obs_keys=None):
        if set(obs_space.keys()) != {'obs'} or len(obs_space['obs']) != 1:
            raise NotImplementedError('the device rollout buffer stores one flat `obs` vector per env')
        self.device = torch.device(device)
        self.num_envs, self.transitions_per_env = int(num_envs), int(horizon_len)
        self.obs_space, self.act_dim = obs_space, int(act_dim)
        self.obs_dim = int(obs_space['obs'][0])
        self.obs_pad = (self.obs_dim + 3) // 4 * 4      # rows are read as 16-byte vectors: zero columns up to a multiple of 4
        self.batch_size, self.minibatch_size = int(batch_size), int(minibatch_size)
        self.length = self.batch_size // self.minibatch_size
        N, T, O, A, dev = self.num_envs, self.transitions_per_env, self.obs_dim, self.act_dim, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.obs_store = torch.zeros((N, T, self.obs_pad), **f32)     # what the kernels read and write
        self.storage_dict = {
            'obses': {'obs': self.obs_store[:, :, :O] if self.obs_pad != O else self.obs_store},
            'actions': torch.zeros((N, T, A), **f32),
            'mu': torch.zeros((N, T, A), **f32),
            'sigma': torch.zeros((N, T, A), **f32),
            'neglogp': torch.zeros((T, N), **f32),
            'values': torch.zeros((T, N, 1), **f32),
            'rewards': torch.zeros((T, N, 1), **f32),
            'dones': torch.zeros((T, N), dtype=torch.uint8, device=dev),
        }
        # flat per-sample scalars produced by compute_return (index n*T + t)
        self._flat = {k: torch.zeros(N * T, **f32) for k in ('values', 'returns', 'advantages', 'neglogp')}
        self.last_values = torch.zeros(N, **f32)
        self.stats = torch.zeros(8, **f32)
        self._sums = torch.zeros(6, dtype=torch.float64, device=dev)
        self._ws = torch.empty(((N + 63) // 64) * 48 + 4096, dtype=torch.uint8, device=dev)
        self.data_dict = None
        self.last_range = None
        s = self.storage_dict
        self.c = _hip.Rollout(N, T, self.obs_store.data_ptr(), s['actions'].data_ptr(), s['mu'].data_ptr(),
                              s['sigma'].data_ptr(), s['neglogp'].data_ptr(), s['values'].data_ptr(), s['rewards'].data_ptr(),
                              s['dones'].data_ptr(), self._flat['values'].data_ptr(), self._flat['returns'].data_ptr(),
                              self._flat['advantages'].data_ptr(), self._flat['neglogp'].data_ptr())

    def __len__(self):
        return self.length

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def write_step(self, t, rewards, dones, time_outs, reward_scale, gamma):
        """Rewards / dones of env step t (ppo.py:153-163).  obs, actions, mu, sigma, neglogp and values of the
        step were already written by the policy kernel (ppo.py:146-148)."""
        d8 = dones.view(torch.uint8) if dones.dtype == torch.bool else dones.to(torch.uint8)
        to8 = None
        if time_outs is not None:
            to8 = time_outs.view(torch.uint8) if time_outs.dtype == torch.bool else time_outs.to(torch.uint8)
        _hip.check(_hip.lib().mrl_rollout_write(self.c, t, _hip.ptr(rewards.contiguous()), _hip.ptr(d8.contiguous()),
                                                _hip.ptr(to8), reward_scale, gamma, self._stream()), 'mrl_rollout_write')

    def compute_return(self, last_values, gamma, tau, value_rms=None, all_reduce=None, world_size=1):
        """experience.py:96-108 fused with prepare_training (:110-121) and the value normalisation of
        ppo.py:175-183 when `value_rms` is given.  With `all_reduce` (a callable summing a device tensor over the
        ranks in place) the advantage / value statistics are those of the envs of ALL ranks."""
        lv = last_values.reshape(-1).contiguous()
        rms = value_rms.c_struct() if value_rms is not None else None
        L, s = _hip.lib(), self._stream()
        if all_reduce is None:
            _hip.check(L.mrl_gae_finish(self.c, _hip.ptr(lv), gamma, tau, rms, _hip.ptr(self.stats), _hip.ptr(self._ws),
                                        self._ws.numel(), s), 'mrl_gae_finish')
            return
        _hip.check(L.mrl_gae_scan(self.c, _hip.ptr(lv), gamma, tau, _hip.ptr(self._sums), _hip.ptr(self._ws), self._ws.numel(), s),
                   'mrl_gae_scan')
        all_reduce(self._sums)
        count = float(self.num_envs * self.transitions_per_env * world_size)
        _hip.check(L.mrl_gae_normalize(self.c, _hip.ptr(self._sums), count, rms, _hip.ptr(self.stats), s), 'mrl_gae_normalize')

    def prepare_training(self):
        """Flattened views, keys as experience.py:110-121 (no copies: the storage is already env-major)."""
        N, T = self.num_envs, self.transitions_per_env
        s = self.storage_dict
        self.data_dict = {
            'obses': {'obs': self.obs_store.view(N * T, self.obs_pad)[:, : self.obs_dim] if self.obs_pad != self.obs_dim
                      else self.obs_store.view(N * T, self.obs_dim)},
            'actions': s['actions'].view(N * T, self.act_dim),
            'mu': s['mu'].view(N * T, self.act_dim),
            'sigma': s['sigma'].view(N * T, self.act_dim),
            'values': self._flat['values'].view(N * T, 1),
            'returns': self._flat['returns'].view(N * T, 1),
            'advantages': self._flat['advantages'],
            'neglogp': self._flat['neglogp'],
            'rewards': s['rewards'].transpose(0, 1).reshape(N * T, 1),
            'dones': s['dones'].transpose(0, 1).reshape(N * T),
        }
        return self.data_dict

    def __getitem__(self, idx):
        start, end = idx * self.minibatch_size, (idx + 1) * self.minibatch_size
        self.last_range = (start, end)
        d = self.data_dict
        return (d['values'][start:end], d['neglogp'][start:end], d['advantages'][start:end], d['mu'][start:end],
                d['sigma'][start:end], d['returns'][start:end], d['actions'][start:end],
                {'obs': d['obses']['obs'][start:end]})

    def update_mu_sigma(self, mu, sigma):
        start, end = self.last_range
        self.data_dict['mu'][start:end] = mu
        self.data_dict['sigma'][start:end] = sigma
