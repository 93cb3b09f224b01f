"""Running mean / std normalisers with fp64 device state (interface of mineral/common/normalizers.py)."""

import torch

from .. import _hip


class Identity:
    """mineral/common/normalizers.py:5-16."""

    def __init__(self, shape=None):
        pass

    def update(self, x):
        return

    def normalize(self, x):
        return x

    def unnormalize(self, x):
        return x

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        return


class RunningMeanStd:
    """Same constructor, state names and update rule as mineral/common/normalizers.py:19-78.

    The state lives on `device` in `dtype`; `update`, `normalize` and `unnormalize` run as HIP kernels
    (batch moments in fp64 accumulators rounded to fp32, Welford merge in the state's precision).
    Only fp64 state is supported by the kernels (what PPO builds at mineral/agents/ppo/ppo.py:30).
    """

    def __init__(self, shape, eps=1e-4, correction=1, with_clamp=False, clamp_range=(-5.0, 5.0), initial_count='eps',
                 dtype=torch.float64, device='cuda'):
        if dtype != torch.float64:
            raise NotImplementedError('the HIP normaliser keeps fp64 state; got %s' % dtype)
        self.shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        if len(self.shape) != 1:
            raise NotImplementedError('only flat feature vectors are normalised on the device path')
        self.eps, self.correction = float(eps), int(correction)
        self.with_clamp, self.clamp_range = with_clamp, tuple(clamp_range)
        if initial_count == 'eps':
            initial_count = eps
        self.device = torch.device(device)
        self.running_mean = torch.zeros(self.shape, dtype=dtype, device=self.device)
        self.running_var = torch.ones(self.shape, dtype=dtype, device=self.device)
        self.running_count = torch.tensor(float(initial_count), dtype=dtype, device=self.device)
        self._moments = torch.empty(2, self.shape[0], dtype=torch.float32, device=self.device)
        self._ws = None

    # -- C struct view ---------------------------------------------------------------------------------------
    def c_struct(self):
        lo, hi = self.clamp_range if self.with_clamp else (-float('inf'), float('inf'))
        return _hip.Rms(self.running_mean.data_ptr(), self.running_var.data_ptr(), self.running_count.data_ptr(),
                        self.shape[0], self.eps, lo, hi)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def to(self, device):
        return self

    def update(self, x):
        """normalizers.py:57-63 for x of shape (rows, size)."""
        x = x.contiguous()
        rows, cols = x.shape
        need = 1024 * 2 * cols * 8 + 65536
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        L = _hip.lib()
        s = self._stream()
        _hip.check(L.mrl_batch_moments(_hip.ptr(x), rows, cols, rows, self.correction, _hip.ptr(self._moments[0]),
                                       _hip.ptr(self._moments[1]), _hip.ptr(self._ws), self._ws.numel(), s), 'mrl_batch_moments')
        st = self.c_struct()
        _hip.check(L.mrl_rms_merge(st, _hip.ptr(self._moments[0]), _hip.ptr(self._moments[1]), float(rows), None, None, s),
                   'mrl_rms_merge')

    def _apply(self, x, unnorm):
        x = x.contiguous()
        y = torch.empty_like(x)
        st = self.c_struct()
        rows = x.numel() // self.shape[0]
        _hip.check(_hip.lib().mrl_rms_normalize(st, _hip.ptr(x), _hip.ptr(y), rows, unnorm, self._stream()), 'mrl_rms_normalize')
        return y

    def normalize(self, x):
        return self._apply(x, 0)

    def unnormalize(self, x):
        return self._apply(x, 1)

    def state_dict(self):
        return {'running_mean': self.running_mean.clone(), 'running_var': self.running_var.clone(),
                'running_count': self.running_count.clone()}

    def load_state_dict(self, sd):
        self.running_mean.copy_(sd['running_mean'])
        self.running_var.copy_(sd['running_var'])
        self.running_count.copy_(sd['running_count'])

    def __repr__(self):
        return f'RunningMeanStd(shape={self.shape})'
