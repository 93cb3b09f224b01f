"""ActorCritic parameter container for the HIP towers (interface of mineral/agents/ppo/models.py:29-151).

All parameters live in ONE flat fp32 device vector whose layout is fixed by the library
(`mrl_param_offsets`); the tensors exposed under the reference's `state_dict` names are views into it, so a
checkpoint written by either implementation loads into the other.
"""

import math
from collections import OrderedDict

import torch

from ... import _hip


class ActorCritic:
    def __init__(self, obs_space, action_dim, mlp_kwargs=None, critic_mlp_kwargs=None, separate_value_mlp=True,
                 fixed_sigma=True, actor_dist_kwargs=None, weight_init='orthogonal', encoder=None, encoder_kwargs=None,
                 device='cuda'):
        mlp_kwargs = dict(mlp_kwargs or dict(units=[512, 256, 128], act_type='ELU'))
        dist = dict(actor_dist_kwargs or dict(dist_type='normal'))
        unsupported = []
        if 'obs' not in obs_space:
            unsupported.append('dict observations (MultiEncoder)')
        if not separate_value_mlp:
            unsupported.append('separate_value_mlp=False')
        if critic_mlp_kwargs is not None:
            unsupported.append('critic_mlp_kwargs')
        kind = (mlp_kwargs.get('norm_type', None), mlp_kwargs.get('act_type', 'ReLU'))
        if kind not in _hip.TOWER_KINDS:
            unsupported.append(f'norm_type={kind[0]} with act_type={kind[1]} (built: no norm or LayerNorm with SiLU, ELU, ReLU or Tanh)')
        if mlp_kwargs.get('dropout', None) is not None or mlp_kwargs.get('plain_last', False) or mlp_kwargs.get('out_dim', None):
            unsupported.append('dropout / plain_last / out_dim')
        if dist.get('dist_type', 'normal') not in _hip.DIST_TYPES:
            unsupported.append(f"dist_type={dist.get('dist_type')}")
        if dist.get('minlogstd', None) is not None or dist.get('maxlogstd', None) is not None:
            unsupported.append('minlogstd / maxlogstd')
        if weight_init != 'orthogonal':
            unsupported.append(f'weight_init={weight_init}')
        if unsupported:
            raise NotImplementedError('unsupported by the MI355X PPO kernels: ' + ', '.join(unsupported))
        self.obs_space, self.action_dim = obs_space, int(action_dim)
        self.obs_dim = int(obs_space['obs'][0])
        # the kernels read observation rows as 16-byte vectors: rows (and the first layer's weight rows) are padded with zero
        # columns to a multiple of 4; the tensors exposed under the reference's names keep the true width
        self.obs_pad = (self.obs_dim + 3) // 4 * 4
        self.units = [int(u) for u in mlp_kwargs.get('units', [512, 256, 128])]
        self.ln_eps = float(mlp_kwargs.get('norm_kwargs', {}).get('eps', 1e-5))
        self.dist_type = dist.get('dist_type', 'normal')
        self.minstd, self.maxstd = float(dist.get('minstd', 1.0)), float(dist.get('maxstd', 1.0))
        self.fixed_sigma, self.separate_value_mlp = bool(fixed_sigma), True
        self.norm_type, self.act_type = kind
        self.device = torch.device(device)
        self.net = _hip.make_net(self.obs_pad, self.action_dim, self.units, self.dist_type, self.minstd, self.maxstd, self.ln_eps,
                                 tower_kind=_hip.TOWER_KINDS[kind], fixed_sigma=self.fixed_sigma)
        L = _hip.lib()
        count = L.mrl_param_count(self.net)
        if count < 0:
            _hip.check(int(count), 'mrl_param_count')
        self.param_count = int(count)
        self.flat = torch.zeros(self.param_count, dtype=torch.float32, device=self.device)
        self._views = self._make_views(self.flat)
        self.training = True
        self.reset_parameters()

    # ---- flat <-> named views ----------------------------------------------------------------------------------
    def _offsets(self):
        import ctypes as C
        offs = (C.c_int64 * (2 * 4 * _hip.MAX_LAYERS + 4))()
        n = C.c_int32(0)
        _hip.check(_hip.lib().mrl_param_offsets(self.net, offs, C.byref(n)), 'mrl_param_offsets')
        return list(offs[: n.value])

    def _make_views(self, flat):
        """Views of `flat` under the reference's parameter names, in the order of its model.parameters()."""
        offs = self._offsets()
        A, U = self.action_dim, self.units[-1]
        has_norm = self.norm_type is not None
        stride = 3 if has_norm else 2           # modules per layer in the reference's nn.Sequential (mlp.py:50-77)
        v = OrderedDict()
        k = 0
        for tower in ('actor_mlp', 'value_mlp'):
            in_f = self.obs_pad
            for i, u in enumerate(self.units):
                b = stride * i
                w = flat[offs[k]: offs[k] + u * in_f].view(u, in_f)
                v[f'{tower}.mlp.{b}.weight'] = w[:, : self.obs_dim] if (i == 0 and self.obs_pad != self.obs_dim) else w
                v[f'{tower}.mlp.{b}.bias'] = flat[offs[k + 1]: offs[k + 1] + u]
                k += 2
                if has_norm:
                    v[f'{tower}.mlp.{b + 1}.weight'] = flat[offs[k]: offs[k] + u]
                    v[f'{tower}.mlp.{b + 1}.bias'] = flat[offs[k + 1]: offs[k + 1] + u]
                    k += 2
                in_f = u
        wh, bh, wv, bv = offs[k: k + 4]
        v['value.weight'] = flat[wv: wv + U].view(1, U)
        v['value.bias'] = flat[bv: bv + 1]
        v['mu.weight'] = flat[wh: wh + A * U].view(A, U)
        v['mu.bias'] = flat[bh: bh + A]
        if self.fixed_sigma:
            # a direct nn.Parameter of the module (models.py:66-67): first in the reference's parameter order
            v['sigma'] = flat[bh + A: bh + 2 * A]
            v.move_to_end('sigma', last=False)
        else:
            v['sigma.weight'] = flat[wh + A * U: wh + 2 * A * U].view(A, U)
            v['sigma.bias'] = flat[bh + A: bh + 2 * A]
        return v

    def views_of(self, flat):
        """Named views into another flat vector of the same layout (gradients, Adam moments)."""
        return self._make_views(flat)

    def named_parameters(self):
        return self._views.items()

    def parameters(self):
        return self._views.values()

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self._views.items())

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._views if k not in sd]
        extra = [k for k in sd if k not in self._views]
        if strict and (missing or extra):
            raise RuntimeError(f'state_dict mismatch: missing {missing}, unexpected {extra}')
        with torch.no_grad():
            for k, v in self._views.items():
                if k in sd:
                    v.copy_(sd[k].to(self.device, torch.float32))

    # ---- init (mineral/agents/ppo/models.py:8-17,75-91) ------------------------------------------------------------
    def reset_parameters(self):
        with torch.no_grad():
            for name, v in self._views.items():
                if name.endswith('.weight') and v.dim() == 2:
                    gain = math.sqrt(2.0)
                    if name == 'value.weight':
                        gain = 1.0
                    elif name in ('mu.weight', 'sigma.weight'):
                        gain = 0.01
                    w = torch.empty(v.shape, dtype=torch.float32)
                    torch.nn.init.orthogonal_(w, gain=gain)
                    v.copy_(w)
                elif name.endswith('.weight'):   # LayerNorm gain
                    v.fill_(1.0)
                else:                            # biases, and the fixed log-std (models.py:87-88)
                    v.zero_()

    # ---- inference API of the reference's module (models.py:93-126), outside the training loop -----------------------
    def _scratch(self, rows):
        """A one-step rollout buffer + workspace for `rows` samples, kept between calls."""
        sc = getattr(self, '_act_scratch', None)
        if sc is None or sc['rows'] != rows:
            dev, A, O = self.device, self.action_dim, self.obs_pad
            f32 = dict(dtype=torch.float32, device=dev)
            t = {k: torch.zeros(rows, 1, A, **f32) for k in ('actions', 'mu', 'sigma')}
            t['obs'] = torch.zeros(rows, 1, O, **f32)
            t.update({k: torch.zeros(1, rows, **f32) for k in ('neglogp', 'values', 'rewards')})
            t['dones'] = torch.zeros(1, rows, dtype=torch.uint8, device=dev)
            t['flat'] = torch.zeros(4, rows, **f32)
            c = _hip.Rollout(rows, 1, t['obs'].data_ptr(), t['actions'].data_ptr(), t['mu'].data_ptr(), t['sigma'].data_ptr(),
                             t['neglogp'].data_ptr(), t['values'].data_ptr(), t['rewards'].data_ptr(), t['dones'].data_ptr(),
                             t['flat'][0].data_ptr(), t['flat'][1].data_ptr(), t['flat'][2].data_ptr(), t['flat'][3].data_ptr())
            ws = torch.empty(int(_hip.lib().mrl_workspace_bytes(self.net, rows)), dtype=torch.uint8, device=dev)
            sc = self._act_scratch = dict(rows=rows, t=t, c=c, ws=ws)
        return sc

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    @torch.no_grad()
    def act(self, obs_dict, sample=True, noise=None):
        """models.py:110-126.  `obs_dict['obs']` is what the reference's module receives (already normalised by the agent).
        sample=False returns the raw output of the mu head (a tensor, not a dict), like the reference.  `noise` (rows, A)
        injects the standard-normal draw; default torch.randn on the device."""
        obs = (obs_dict['obs'] if isinstance(obs_dict, dict) else obs_dict).to(self.device, torch.float32)
        if self.obs_pad != self.obs_dim:
            obs = torch.nn.functional.pad(obs, (0, self.obs_pad - self.obs_dim))
        obs = obs.contiguous()
        rows = obs.shape[0]
        sc = self._scratch(rows)
        L = _hip.lib()
        if not sample:
            mu = torch.empty(rows, self.action_dim, dtype=torch.float32, device=self.device)
            _hip.check(L.mrl_policy_mean(self.net, _hip.ptr(self.flat), None, _hip.ptr(obs), rows, _hip.ptr(mu), _hip.ptr(sc['ws']),
                                         sc['ws'].numel(), self._stream()), 'mrl_policy_mean')
            return mu
        if noise is None:
            noise = torch.randn(rows, self.action_dim, dtype=torch.float32, device=self.device)
        noise = noise.to(self.device, torch.float32).contiguous()
        _hip.check(L.mrl_policy_act(self.net, _hip.ptr(self.flat), None, None, _hip.ptr(obs), _hip.ptr(noise), 0, sc['c'], None, None,
                                    _hip.ptr(sc['ws']), sc['ws'].numel(), self._stream()), 'mrl_policy_act')
        t = sc['t']
        mu, sigma = t['mu'][:, 0].clone(), t['sigma'][:, 0].clone()
        return {'neglogp': t['neglogp'][0].clone(), 'values': t['values'][0].clone().unsqueeze(1), 'actions': t['actions'][:, 0].clone(),
                'mu': mu, 'sigma': sigma, 'distr': torch.distributions.Normal(mu, sigma)}

    @torch.no_grad()
    def forward(self, input_dict):
        """models.py:93-108 as an inference call: values, entropy, mu, sigma (and prev_neglogp when `prev_actions` is given) from
        the device towers.  No autograd graph is built: the training loop gets its gradients from mrl_ppo_grad."""
        out = self.act(input_dict, sample=True, noise=torch.zeros(input_dict['obs'].shape[0], self.action_dim, device=self.device))
        distr = out['distr']
        res = {'values': out['values'], 'entropy': distr.entropy().sum(dim=-1), 'mu': out['mu'], 'sigma': out['sigma'], 'distr': distr}
        prev = input_dict.get('prev_actions', None)
        if prev is not None:
            res['prev_neglogp'] = torch.squeeze(-distr.log_prob(prev.to(self.device)).sum(dim=-1))
        return res

    __call__ = forward

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        return self

    def __repr__(self):
        tower = f'LayerNorm+{self.act_type}' if self.norm_type else self.act_type
        return (f'ActorCritic[HIP gfx950](obs={self.obs_dim}, act={self.action_dim}, units={self.units}, {tower}, '
                f'dist={self.dist_type}{", fixed sigma" if self.fixed_sigma else ""}, params={self.param_count})')
