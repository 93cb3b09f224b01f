"""Device critic regression of SHAC: the ensemble critic, TD(lambda) targets, the critic dataset, `update_critic` and the
target blend, with the interfaces of mineral/agents/diffrl/models.py:181-256, utils.py:25-69 and shac.py:824-889.

Every numeric step runs in the HIP library (include/mineral_shac.h, mrl_adam_step); there is no CPU fallback."""

import collections
import ctypes as C

import numpy as np
import torch

from ... import _hip
from ..sac.models import _Flat, _mlp_kwargs_ok, _tower_parts


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class EnsembleCritic(_Flat):
    """models.py:181-256 with with_vmap=False, n_sample=None: n_critics towers on the observation, each ending in a plain
    Linear to 1.  One flat fp32 device vector in the library's layout; `state_dict` uses the reference's names."""

    DEFAULT_MLP_KWARGS = dict(act_type='ELU', norm_type='LayerNorm')   # models.py:154,189

    def __init__(self, state_dim, action_dim=None, n_critics=1, n_sample=None, with_vmap=False, mlp_kwargs=None, weight_init='orthogonal', device='cuda'):
        if n_critics not in (1, 2) or n_sample is not None or with_vmap:
            raise NotImplementedError('the HIP critic ensemble has n_critics in {1, 2}, n_sample=None, with_vmap=False')
        self._setup(state_dim, [f'critics.{c}' for c in range(n_critics)], mlp_kwargs, weight_init, device)

    def _setup(self, state_dim, prefixes, mlp_kwargs, weight_init, device):
        _hip.lib()
        kw = dict(self.DEFAULT_MLP_KWARGS if mlp_kwargs is None else mlp_kwargs)
        units = _mlp_kwargs_ok(kw, type(self).__name__, acts=tuple(_hip.CRITIC_ACTS))
        if weight_init not in ('orthogonal', 'orthogonalg1'):
            raise NotImplementedError(f'weight_init={weight_init}: the HIP critics cover orthogonal and orthogonalg1')
        if not str(device).startswith('cuda'):
            raise _hip.HipLibraryError(f'the critic runs on an MI355X only (device={device}); no CPU fallback')
        self.act_type = kw['act_type']
        self.net = _hip.make_critic_net(int(state_dim), len(prefixes), units, act_type=self.act_type)
        parts = []
        for pre in prefixes:
            parts += _tower_parts(pre, state_dim, units)
        b = 3 * len(units)
        for pre in prefixes:
            parts += [[(f'{pre}.mlp.{b}.weight', (1, units[-1]))], [(f'{pre}.mlp.{b}.bias', (1,))]]
        super().__init__(self.net, None, parts, device)
        self.units, self.n_critics, self.state_dim = units, len(prefixes), int(state_dim)
        self._prefixes = list(prefixes)
        self.weight_init = weight_init
        self._ws = None
        self.reset_parameters()

    def reset_parameters(self):
        """models.py:10-14,72-78,226-233: orthogonal Linear weights with gain sqrt(2) ('orthogonal') or 1 ('orthogonalg1'), zero
        biases; LayerNorm keeps torch's default (ones / zeros)."""
        gain = 1.4142135623730951 if self.weight_init == 'orthogonal' else 1.0
        with torch.no_grad():
            for name, v in self._views.items():
                if name.endswith('.weight') and v.dim() == 2:
                    w = torch.empty(v.shape, dtype=torch.float32)
                    torch.nn.init.orthogonal_(w, gain=gain)
                    v.copy_(w)
                elif name.endswith('.weight'):
                    v.fill_(1.0)
                else:
                    v.zero_()

    @staticmethod
    def _query(net, which):
        L = _hip.lib()
        count = int(L.mrl_critic_param_count(net))
        if count < 0:
            _hip.check(count, 'mrl_critic_param_count')
        offs = (C.c_int64 * (2 * 4 * _hip.MAX_LAYERS + 8))()
        n = C.c_int32(0)
        _hip.check(L.mrl_critic_param_offsets(net, offs, C.byref(n)), 'mrl_critic_param_offsets')
        return count, list(offs)[: n.value]

    def _ordered(self):
        keys = []
        for pre in self._prefixes:
            keys += [k for k in self._views if k.startswith(pre + '.')]
        return collections.OrderedDict((k, self._views[k]) for k in keys)

    def workspace(self, rows):
        need = int(_hip.lib().mrl_critic_workspace_bytes(self.net, rows))
        if need < 0:
            _hip.check(need, 'mrl_critic_workspace_bytes')
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _values(self, x):
        if isinstance(x, dict):
            x = x['z']
        lead = x.shape[:-1]
        flat = x.reshape(-1, self.state_dim).contiguous().float()
        rows = flat.shape[0]
        out = torch.empty(self.n_critics, rows, dtype=torch.float32, device=self.device)
        ws = self.workspace(rows)
        _hip.check(_hip.lib().mrl_critic_values(self.net, _hip.ptr(self.flat), _hip.ptr(flat), rows, None, None, 0.0, 0.0, _hip.ptr(out), _hip.ptr(ws),
                                                ws.numel(), _stream()), 'mrl_critic_values')
        return out.view(self.n_critics, *lead, 1)

    def forward(self, x, return_type='min'):
        """models.py:235-256.  x (..., state_dim) fp32 (or {'z': x}); values come back shaped (..., 1) per critic."""
        Vs = self._values(x)
        if return_type == 'all':
            return Vs
        if return_type == 'min':
            return torch.min(Vs, dim=0).values
        if return_type == 'min_and_avg':
            return torch.min(Vs, dim=0).values, torch.mean(Vs, dim=0)
        return Vs

    __call__ = forward

    def clone(self):
        """deepcopy for the target critic (shac.py:166)."""
        other = object.__new__(type(self))
        other.__dict__.update(self.__dict__)
        other.flat = self.flat.clone()
        other._views = other.views_of(other.flat)
        other._ws = None
        return other

    def __repr__(self):
        return f'{type(self).__name__}[HIP gfx950](n_critics={self.n_critics}, units={self.units}, LayerNorm+{self.act_type}, params={self.param_count})'


class Critic(EnsembleCritic):
    """models.py:149-178: one tower under the name `critic_mlp` (the critic of DFlexAntSHAC.yaml, LayerNorm + ELU)."""

    def __init__(self, state_dim, action_dim=None, mlp_kwargs=None, weight_init='orthogonal', device='cuda'):
        self._setup(state_dim, ['critic_mlp'], mlp_kwargs, weight_init, device)

    def forward(self, x, return_type=None):
        v = self._values(x)[0]
        if return_type == 'all':
            return [v]
        if return_type == 'min_and_avg':
            return v, v.clone()
        return v

    __call__ = forward


def soft_update(module, module_target, alpha):
    """utils.py:25-28: target <- alpha * target + (1 - alpha) * current."""
    n = module.flat.numel()
    if module_target.flat.numel() != n:
        raise ValueError('soft_update: the two critics differ in size')
    _hip.check(_hip.lib().mrl_lerp_params(_hip.ptr(module_target.flat), _hip.ptr(module.flat), n, float(alpha), float(1.0 - alpha), _stream()),
               'mrl_lerp_params')


def grad_norm(grads):
    """utils.py:31-36 for a flat gradient vector."""
    return torch.sqrt(torch.sum(grads ** 2))


def compute_target_values(rew_buf, done_mask, next_values, gamma, lam=0.95, critic_method='td-lambda', out=None):
    """shac.py:877-891 on (horizon, num_envs) fp32 device buffers."""
    if critic_method not in ('one-step', 'td-lambda'):
        raise NotImplementedError(critic_method)
    T, N = rew_buf.shape
    for t in (rew_buf, done_mask, next_values):
        if t.shape != (T, N) or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError('compute_target_values: buffers must be contiguous fp32 (horizon, num_envs) device tensors')
    if out is None:
        out = torch.empty_like(rew_buf)
    _hip.check(_hip.lib().mrl_tdlambda_targets(_hip.ptr(rew_buf), _hip.ptr(done_mask), _hip.ptr(next_values), N, T, float(gamma), float(lam),
                                               1 if critic_method == 'one-step' else 0, _hip.ptr(out), _stream()), 'mrl_tdlambda_targets')
    return out


def compute_target_values_with_entropy(rew_buf, done_mask, next_values, entropy, alpha, gamma, lam=0.95, critic_method='td-lambda',
                                       offset_by_target_entropy=False, scale_by_target_entropy=False, target_entropy=None, out=None):
    """shac.py:893-916.  `entropy` is the distribution entropy or -logprobs, (horizon, num_envs); `alpha` the entropy
    coefficient or the current temperature as a Python float."""
    if critic_method not in ('one-step', 'td-lambda'):
        raise NotImplementedError(critic_method)
    T, N = rew_buf.shape
    if offset_by_target_entropy:
        entropy = (entropy + abs(target_entropy)) * 0.5
    if scale_by_target_entropy:
        entropy = entropy * (1.0 / abs(target_entropy))
    entropy = entropy.contiguous()
    for t in (rew_buf, done_mask, next_values, entropy):
        if t.shape != (T, N) or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError('compute_target_values_with_entropy: buffers must be contiguous fp32 (horizon, num_envs) device tensors')
    if out is None:
        out = torch.empty_like(rew_buf)
    _hip.check(_hip.lib().mrl_tdlambda_targets_entropy(_hip.ptr(rew_buf), _hip.ptr(done_mask), _hip.ptr(next_values), _hip.ptr(entropy), float(alpha), N, T,
                                                       float(gamma), float(lam), 1 if critic_method == 'one-step' else 0, _hip.ptr(out), _stream()),
               'mrl_tdlambda_targets_entropy')
    return out


class CriticDataset:
    """utils.py:39-69: (T, B, ...) buffers flattened time-major into contiguous batches, the last one short unless drop_last."""

    def __init__(self, batch_size, obs, target_values, shuffle=False, drop_last=False):
        self.obs = {k: v.reshape(-1, *v.shape[2:]) for k, v in obs.items()}
        self.target_values = target_values.reshape(-1)
        self.N = self.target_values.shape[0]
        self.batch_size = batch_size
        if shuffle:
            self.shuffle()
        self.length = self.N // batch_size if drop_last else ((self.N - 1) // batch_size) + 1

    def shuffle(self):
        index = torch.from_numpy(np.random.permutation(self.N)).to(self.target_values.device)
        self.obs = {k: v[index, ...] for k, v in self.obs.items()}
        self.target_values = self.target_values[index]

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        s, e = index * self.batch_size, min((index + 1) * self.batch_size, self.N)
        return {k: v[s:e, ...] for k, v in self.obs.items()}, self.target_values[s:e]


class CriticLearner:
    """The critic half of the SHAC agent: optimizer state (shac.py:144-156), `update_critic` (shac.py:824-867),
    `compute_critic_loss` (shac.py:869-875) and the target blend (shac.py:429-434)."""

    def __init__(self, critic, optim_type='AdamW', critic_optim_kwargs=None, critic_iterations=16, max_grad_norm=0.5, truncate_grads=True,
                 target_critic_alpha=0.4, no_target_critic=False, obs_key='obs'):
        kw = dict(critic_optim_kwargs or {})
        if optim_type not in ('Adam', 'AdamW'):
            raise NotImplementedError(f'optim_type={optim_type}')
        self.critic = critic
        self.critic_target = critic if no_target_critic else critic.clone()
        self.no_target_critic = no_target_critic
        self.adamw = 1 if optim_type == 'AdamW' else 0
        self.lr = float(kw.get('lr', 1e-3))
        self.betas = tuple(kw.get('betas', (0.9, 0.999)))
        self.eps = float(kw.get('eps', 1e-8))
        self.weight_decay = float(kw.get('weight_decay', 0.01 if self.adamw else 0.0))
        self.critic_iterations, self.max_grad_norm, self.truncate_grads = int(critic_iterations), max_grad_norm, truncate_grads
        self.target_critic_alpha, self.obs_key = target_critic_alpha, obs_key
        n, dev = critic.flat.numel(), critic.device
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.scratch = torch.empty(4096, dtype=torch.uint8, device=dev)
        self.metrics = torch.zeros(4, dtype=torch.float64, device=dev)
        self.step = 0

    def _obs(self, b_obs):
        x = b_obs[self.obs_key] if isinstance(b_obs, dict) else b_obs
        return x.contiguous()

    def _loss_and_grad(self, b_obs, b_target):
        c = self.critic
        x, tgt = self._obs(b_obs), b_target.contiguous()
        rows = tgt.shape[0]
        if x.shape != (rows, c.state_dim):
            raise ValueError(f'critic batch: obs {tuple(x.shape)} vs targets {tuple(tgt.shape)}')
        ws = c.workspace(rows)
        _hip.check(_hip.lib().mrl_critic_mse_grad(c.net, _hip.ptr(c.flat), _hip.ptr(x), _hip.ptr(tgt), rows, None, None, 0.0, 0.0, _hip.ptr(self.grads),
                                                  _hip.ptr(self.metrics), _hip.ptr(ws), ws.numel(), _stream()), 'mrl_critic_mse_grad')
        return self.metrics[0].clone()

    def compute_critic_loss(self, obs, target_v):
        """shac.py:869-875 (mean=True, reduction='mean'); leaves the gradient of the loss in `self.grads`."""
        return self._loss_and_grad(obs, target_v).float()

    def update_critic(self, dataset):
        results = collections.defaultdict(list)
        L = _hip.lib()
        p = self.critic.flat
        clip = self.truncate_grads and self.max_grad_norm is not None
        for _ in range(self.critic_iterations):
            total, before, after = 0.0, [], []
            B = len(dataset)
            for i in range(B):
                b_obs, b_target = dataset[i]
                loss = self._loss_and_grad(b_obs, b_target)
                self.step += 1
                _hip.check(L.mrl_adam_step(_hip.ptr(p), _hip.ptr(self.grads), _hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq), p.numel(), self.lr,
                                           self.betas[0], self.betas[1], self.eps, self.weight_decay, self.adamw, self.step,
                                           float(self.max_grad_norm) if clip else 0.0, 1.0, _hip.ptr(self.gnorm), _hip.ptr(self.scratch), _stream()),
                           'mrl_adam_step')
                if self.truncate_grads:
                    g = self.gnorm[0].clone()
                    before.append(g)
                    # clip_grad_norm_ scales by min(1, max_norm / (norm + 1e-6)); the norm after it is the product
                    after.append(g * torch.clamp(self.max_grad_norm / (g + 1e-6), max=1.0) if clip else g)
                total = total + loss
            results['value_loss'].append((total / B).float())
            if self.truncate_grads:
                results['grad_norm_before_clip/critic'].append(torch.mean(torch.stack(before)))
                results['grad_norm_after_clip/critic'].append(torch.mean(torch.stack(after)))
        return results

    def set_lr(self, lr):
        """The learning-rate schedules of shac.py:363-380 write the optimizer's lr each epoch."""
        self.lr = float(lr)

    def update_target(self, alpha=None):
        if not self.no_target_critic:
            soft_update(self.critic, self.critic_target, self.target_critic_alpha if alpha is None else alpha)
