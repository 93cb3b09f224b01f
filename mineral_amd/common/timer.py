"""Wall-clock scopes with the statistics of mineral/common/timer.py, plus an optional device sync at scope
ends so GPU work is attributed to the scope that enqueued it."""

import collections
import contextlib
import time

import numpy as np


class Timer:
    COLUMNS = ('count', 'sum', 'frac', 'avg', 'min', 'max', 'total', 'totalrate', 'lastrate')

    def __init__(self, columns=COLUMNS, sync=None):
        assert all(c in self.COLUMNS for c in columns), columns
        self.columns = columns
        self._sync = sync
        self._durations = collections.defaultdict(list)
        self._totals = collections.defaultdict(float)
        self._open = {}
        self._last = time.perf_counter()
        self._step = 0

    @contextlib.contextmanager
    def scope(self, name, sync=True):
        """sync=False times the host side only, like the reference: for scopes entered once per env step, where a device
        sync at both ends would stop the host from running ahead of the device."""
        if self._sync and sync:
            self._sync()
        t0 = time.perf_counter()
        yield
        if self._sync and sync:
            self._sync()
        self._durations[name].append(time.perf_counter() - t0)

    def wrap(self, name, obj, methods, sync=True):
        for method in methods:
            setattr(obj, method, self.scope(f'{name}.{method}', sync=sync)(getattr(obj, method)))

    def start(self, name):
        if name in self._open:
            raise ValueError(f"Scope '{name}' is already active")
        self._open[name] = self.scope(name)
        self._open[name].__enter__()

    def end(self, name):
        if name not in self._open:
            raise ValueError(f"Scope '{name}' is not active.")
        self._open.pop(name).__exit__(None, None, None)

    def reset(self):
        for cm in self._open.values():
            cm.__exit__(None, None, None)
        self._open.clear()
        for d in self._durations.values():
            d.clear()
        self._last = time.perf_counter()

    def stats(self, step=None, reset=True, total_names=None):
        out = {'duration': time.perf_counter() - self._last}
        for name, durs in self._durations.items():
            tot = float(np.sum(durs))
            row = {'count': len(durs), 'sum': tot, 'frac': tot / out['duration']}
            if durs:
                row.update(avg=float(np.mean(durs)), min=float(np.min(durs)), max=float(np.max(durs)))
            self._totals[name] += tot
            row['total'] = self._totals[name]
            if step is not None:
                row['totalrate'] = step / row['total'] if row['total'] else 0.0
                row['lastrate'] = (step - self._step) / tot if tot else 0.0
            out.update({f'{name}/{k}': v for k, v in row.items() if k in self.columns})
        if total_names is not None:
            out['total'] = sum(self._totals[k] for k in total_names)
            if step is not None:
                out['totalrate'] = step / out['total'] if out['total'] else 0.0
                out['lastrate'] = (step - self._step) / out['duration']
        if reset:
            if step is not None:
                self._step = step
            self.reset()
        else:
            self._totals.clear()
        return out
