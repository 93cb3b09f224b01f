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
                 dtype=torch.float64, device='cuda', pad_to=None):
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
        # `pad_to`: the kernels read feature vectors whose length is a multiple of 4; the state then lives in vectors of that
        # length (padding features: mean 0, variance 1, and they only ever see zeros) and the reference's names are views
        self.size = int(pad_to) if pad_to else self.shape[0]
        if self.size < self.shape[0]:
            raise ValueError(f'pad_to {pad_to} < {self.shape[0]} features')
        self._mean_store = torch.zeros(self.size, dtype=dtype, device=self.device)
        self._var_store = torch.ones(self.size, dtype=dtype, device=self.device)
        self.running_mean = self._mean_store[: self.shape[0]]
        self.running_var = self._var_store[: self.shape[0]]
        self.running_count = torch.tensor(float(initial_count), dtype=dtype, device=self.device)
        self._moments = torch.empty(2, self.size, dtype=torch.float32, device=self.device)
        self._ws = None

    def _padded(self, x):
        """(rows, features) -> (rows, size) with zero columns appended."""
        x = x.reshape(-1, self.shape[0])
        return x.contiguous() if self.size == self.shape[0] else torch.nn.functional.pad(x, (0, self.size - self.shape[0])).contiguous()

    # -- C struct view ---------------------------------------------------------------------------------------
    def c_struct(self):
        lo, hi = self.clamp_range if self.with_clamp else (-float('inf'), float('inf'))
        return _hip.Rms(self._mean_store.data_ptr(), self._var_store.data_ptr(), self.running_count.data_ptr(),
                        self.size, self.eps, lo, hi)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def to(self, device):
        return self

    def update(self, x):
        """normalizers.py:57-63 for x of shape (rows, size)."""
        x = self._padded(x)
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
        shape = x.shape
        x = self._padded(x)
        y = torch.empty_like(x)
        st = self.c_struct()
        _hip.check(_hip.lib().mrl_rms_normalize(st, _hip.ptr(x), _hip.ptr(y), x.shape[0], unnorm, self._stream()), 'mrl_rms_normalize')
        return y[:, : self.shape[0]].reshape(shape) if self.size != self.shape[0] else y.reshape(shape)

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
