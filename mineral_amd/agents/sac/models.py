"""Actor and EnsembleQ parameter containers for the HIP SAC path (interfaces of mineral/agents/ddpg/models.py:8-118).

Each network is ONE flat fp32 device vector in the library's layout (`mrl_sac_param_offsets`); the tensors exposed
under the reference's `state_dict` names are views into it, so checkpoints interchange."""

import ctypes as C
from collections import OrderedDict

import torch

from ... import _hip


def tower_kind_of(kw, what):
    """MRL_TOWER_* of an mlp_kwargs dict (norm_type None or LayerNorm; act_type SiLU, ELU, ReLU or Tanh; mineral/nets/mlp.py:5-22)."""
    kind = (kw.get('norm_type', None), kw.get('act_type', 'ReLU'))
    bad = []
    if kind not in _hip.TOWER_KINDS:
        bad.append(f'norm_type={kind[0]} with act_type={kind[1]}')
    if kw.get('dropout', None) is not None:
        bad.append('dropout')
    if bad:
        raise NotImplementedError(f'{what}: the MI355X kernels cover towers without a norm layer or with LayerNorm, with SiLU, ELU, ReLU or Tanh; '
                                  'unsupported here: ' + ', '.join(bad))
    return _hip.TOWER_KINDS[kind]


def _mlp_kwargs_ok(kw, what, acts=None):
    """units (and whether the towers have a norm layer) of an mlp_kwargs dict.  acts: callers whose kernels only take LayerNorm
    towers with one of these activations (the SHAC critics) get the units alone."""
    if acts is not None:
        bad = []
        if kw.get('norm_type', None) != 'LayerNorm':
            bad.append(f"norm_type={kw.get('norm_type', None)}")
        if kw.get('act_type', 'ReLU') not in acts:
            bad.append(f"act_type={kw.get('act_type', 'ReLU')}")
        if kw.get('dropout', None) is not None:
            bad.append('dropout')
        if bad:
            raise NotImplementedError(f"{what}: the MI355X kernels cover LayerNorm+{'/'.join(acts)} towers; unsupported here: " + ', '.join(bad))
        return [int(u) for u in kw.get('units', [512, 256, 128])]
    tower_kind_of(kw, what)
    return [int(u) for u in kw.get('units', [512, 256, 128])], kw.get('norm_type', None) is not None


class _Flat:
    """Named views over a flat parameter vector."""

    def __init__(self, net, which, names_shapes, device):
        count, offs = self._query(net, which)
        self.param_count = count
        self.device = torch.device(device)
        self._layout = list(zip(names_shapes, offs))   # ((names, shapes) per library tensor, offset)
        self.flat = torch.zeros(count, dtype=torch.float32, device=self.device)
        self._views = self.views_of(self.flat)
        self.training = True

    @staticmethod
    def _query(net, which):
        """(parameter count, offsets of the library's tensors) of network `which` of an mrl_sac_net."""
        L = _hip.lib()
        count = int(L.mrl_sac_param_count(net, which))
        if count < 0:
            _hip.check(count, 'mrl_sac_param_count')
        offs = (C.c_int64 * (2 * 4 * _hip.MAX_LAYERS + 8))()
        n = C.c_int32(0)
        _hip.check(L.mrl_sac_param_offsets(net, which, offs, C.byref(n)), 'mrl_sac_param_offsets')
        return count, list(offs)[: n.value]

    def views_of(self, flat):
        v = OrderedDict()
        for (parts, off) in self._layout:
            o = off
            for name, shape in parts:   # a library tensor may hold several reference tensors back to back (the [mu; sigma] head)
                n = 1
                for d in shape:
                    n *= d
                v[name] = flat[o:o + n].view(*shape)
                o += n
        return v

    def named_parameters(self):
        return self._views.items()

    def parameters(self):
        return self._views.values()

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self._ordered().items())

    def _ordered(self):
        return self._views

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._views if k not in sd]
        extra = [k for k in sd if k not in self._views]
        if strict and (missing or extra):
            raise RuntimeError(f'state_dict mismatch: missing {missing}, unexpected {extra}')
        with torch.no_grad():
            for k, v in self._views.items():
                if k in sd:
                    v.copy_(sd[k].to(self.device, torch.float32))

    def reset_parameters(self):
        """weight_init='orthogonal' (mineral/agents/ddpg/utils.py:45-49): orthogonal Linear weights (gain 1), zero biases;
        LayerNorm keeps torch's default (ones / zeros)."""
        with torch.no_grad():
            for name, v in self._views.items():
                if name.endswith('.weight') and v.dim() == 2:
                    w = torch.empty(v.shape, dtype=torch.float32)
                    torch.nn.init.orthogonal_(w)
                    v.copy_(w)
                elif name.endswith('.weight'):
                    v.fill_(1.0)
                else:
                    v.zero_()

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        return self

    def requires_grad_(self, flag=True):
        return self


def _tower_parts(prefix, in_dim, units, has_norm=True):
    """Module indices as nn.Sequential numbers them in mineral/nets/mlp.py:50-77: Linear, (norm,) activation per layer."""
    parts, in_f, stride = [], in_dim, 3 if has_norm else 2
    for i, u in enumerate(units):
        b = stride * i
        parts += [[(f'{prefix}.mlp.{b}.weight', (u, in_f))], [(f'{prefix}.mlp.{b}.bias', (u,))]]
        if has_norm:
            parts += [[(f'{prefix}.mlp.{b + 1}.weight', (u,))], [(f'{prefix}.mlp.{b + 1}.bias', (u,))]]
        in_f = u
    return parts


class Actor(_Flat):
    """ddpg/models.py:8-69 with tanh_policy=False, fixed_sigma=False and a squashed_normal Dist."""

    def __init__(self, net, state_dim, action_dim, tanh_policy=True, fixed_sigma=None, mlp_kwargs=None, dist_kwargs=None, weight_init=None, device='cuda'):
        units, has_norm = _mlp_kwargs_ok(dict(mlp_kwargs or {}), 'Actor')
        self.kind_name = f"{'LayerNorm+' if has_norm else ''}{dict(mlp_kwargs or {}).get('act_type', 'ReLU')}"
        dist = dict(dist_kwargs or {})
        if tanh_policy or fixed_sigma is not False or dist.get('dist_type') != 'squashed_normal' or weight_init != 'orthogonal':
            raise NotImplementedError('the HIP SAC actor is tanh_policy=False, fixed_sigma=False, dist_type=squashed_normal, weight_init=orthogonal')
        U = units[-1]
        parts = _tower_parts('actor_mlp', state_dim, units, has_norm)
        parts += [[('mu.weight', (action_dim, U)), ('sigma.weight', (action_dim, U))], [('mu.bias', (action_dim,)), ('sigma.bias', (action_dim,))]]
        super().__init__(net, 0, parts, device)
        self.units = units
        self.reset_parameters()

    def _ordered(self):
        # reference order: tower, mu.weight, mu.bias, sigma.weight, sigma.bias
        keys = [k for k in self._views if k.startswith('actor_mlp')] + ['mu.weight', 'mu.bias', 'sigma.weight', 'sigma.bias']
        return OrderedDict((k, self._views[k]) for k in keys)

    def __repr__(self):
        return f'Actor[HIP gfx950](units={self.units}, {self.kind_name}, squashed_normal, params={self.param_count})'


class EnsembleQ(_Flat):
    """ddpg/models.py:72-118 with n_critics=2: every critic an MLP on cat(state, action) ending in a plain Linear to 1."""

    def __init__(self, net, state_dim, action_dim, n_critics=2, mlp_kwargs=None, weight_init=None, device='cuda'):
        units, has_norm = _mlp_kwargs_ok(dict(mlp_kwargs or {}), 'EnsembleQ')
        self.kind_name = f"{'LayerNorm+' if has_norm else ''}{dict(mlp_kwargs or {}).get('act_type', 'ReLU')}"
        if n_critics != 2 or weight_init != 'orthogonal':
            raise NotImplementedError('the HIP SAC critic ensemble has n_critics=2 and weight_init=orthogonal')
        parts = []
        for c in range(2):
            parts += _tower_parts(f'critics.{c}', state_dim + action_dim, units, has_norm)
        b = (3 if has_norm else 2) * len(units)
        for c in range(2):
            parts += [[(f'critics.{c}.mlp.{b}.weight', (1, units[-1]))], [(f'critics.{c}.mlp.{b}.bias', (1,))]]
        super().__init__(net, 1, parts, device)
        self.units, self.n_critics = units, 2
        self.reset_parameters()

    def _ordered(self):
        keys = []
        for c in range(2):
            keys += [k for k in self._views if k.startswith(f'critics.{c}.')]
        return OrderedDict((k, self._views[k]) for k in keys)

    def clone(self):
        other = object.__new__(EnsembleQ)
        other.__dict__.update(self.__dict__)
        other.flat = self.flat.clone()
        other._views = other.views_of(other.flat)
        return other

    def __repr__(self):
        return f'EnsembleQ[HIP gfx950](n_critics=2, units={self.units}, {self.kind_name}, params={self.param_count})'
