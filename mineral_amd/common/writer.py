"""Metric fan-out with the call pattern of mineral/common/writer.py; sinks whose packages are missing degrade
to no-ops instead of failing the run."""

import json
import os

import numpy as np


class Writer:
    def __init__(self, outputs):
        self.outputs = outputs
        self._metrics = []

    def add(self, step, metrics, prefix=None):
        for k, v in metrics.items():
            self._metrics.append((step, f'{prefix}_{k}' if prefix else k, v))

    def write(self):
        if not self._metrics:
            return
        for output in self.outputs:
            output(tuple(self._metrics))
        self._metrics.clear()


class AsyncOutput:
    """writer.py:24-39: the sink runs on one worker thread; a new batch first waits for the previous one."""

    def __init__(self, callback, parallel=True):
        self._callback = callback
        self._parallel = parallel
        if parallel:
            import concurrent.futures
            self._executor = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            self._future = None

    def __call__(self, summaries):
        if self._parallel:
            self._future and self._future.result()
            self._future = self._executor.submit(self._callback, summaries)
        else:
            self._callback(summaries)

    def flush(self):
        if self._parallel and self._future is not None:
            self._future.result()


def flatten_dict(d, prefix='', sep='.'):
    out = {}
    for k, v in (d or {}).items():
        key = f'{prefix}{sep}{k}' if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten_dict(v, key, sep))
        else:
            out[key] = v
    return out


class TensorboardWriter(AsyncOutput):
    """writer.py:42-82: scalars, histograms (1-D), point clouds ((..., 3) with 3 dims), the flattened config as a markdown table
    under `cfg`.  Without the tensorboard package the sink is a no-op; video summaries (5-D) are refused."""

    def __init__(self, tb_dir, config=None, parallel=True):
        super().__init__(self._write, parallel)
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(tb_dir)
        except Exception:
            self.writer = None
        if self.writer is not None and config:
            try:
                import pandas as pd
                import tabulate  # noqa: F401  (pandas.to_markdown needs it)
                self.writer.add_text('cfg', pd.DataFrame.from_dict(flatten_dict(config), orient='index').to_markdown(), 0)
            except ImportError:
                pass

    def _write(self, summaries):
        if self.writer is None:
            return
        for step, name, value in summaries:
            value = np.asarray(value)
            if value.ndim == 0:
                self.writer.add_scalar(name, value, step)
            elif value.ndim == 1:
                self.writer.add_histogram(name, value, step)
            elif value.ndim == 3 and value.shape[-1] == 3:
                self.writer.add_mesh(name, value, global_step=step)
            elif value.ndim == 5:
                raise NotImplementedError(f'video summary `{name}`: gif encoding (writer.py:84-121) is outside this build')


class WandbWriter:
    def __init__(self):
        try:
            import wandb
            self.wandb = wandb if getattr(wandb, 'run', None) is not None else None
        except Exception:
            self.wandb = None

    def __call__(self, summaries):
        if self.wandb is None:
            return
        bystep = {}
        for step, name, value in summaries:
            value = np.asarray(value)
            if value.ndim == 0:
                bystep.setdefault(step, {})[name] = value.item()
        for step, m in bystep.items():
            self.wandb.log(m, step=step)


class JsonlWriter:
    """Scalar metrics as one JSON line per step under logdir (always available)."""

    def __init__(self, path):
        self.path = path
        os.makedirs(os.path.dirname(path), exist_ok=True)

    def __call__(self, summaries):
        bystep = {}
        for step, name, value in summaries:
            value = np.asarray(value)
            if value.ndim == 0:
                bystep.setdefault(int(step), {})[name] = float(value)
        with open(self.path, 'a') as f:
            for step, m in bystep.items():
                f.write(json.dumps({'step': step, **m}) + '\n')
