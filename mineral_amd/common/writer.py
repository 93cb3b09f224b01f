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


class TensorboardWriter:
    def __init__(self, tb_dir, config=None):
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(tb_dir)
        except Exception:
            self.writer = None

    def __call__(self, summaries):
        if self.writer is None:
            return
        for step, name, value in summaries:
            value = np.asarray(value)
            if value.ndim == 0:
                self.writer.add_scalar(name, value, step)
            elif value.ndim == 1:
                self.writer.add_histogram(name, value, step)


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
