"""Fixed-length window of finished-episode statistics (interface of mineral/common/tracker.py)."""

from collections import deque

import numpy as np


class Tracker:
    def __init__(self, max_len):
        self.max_len = max_len
        self.reset()

    def reset(self):
        self.window = deque([0] * self.max_len, maxlen=self.max_len)

    def update(self, value):
        if hasattr(value, 'tolist'):
            value = value.tolist()
        if isinstance(value, (list, tuple)):
            self.window.extend(value)
        else:
            self.window.append(value)

    def mean(self):
        return float(np.mean(self.window))

    def std(self):
        return float(np.std(self.window))

    def max(self):
        return float(np.max(self.window))

    def min(self):
        return float(np.min(self.window))

    def median(self):
        return float(np.median(self.window))

    def sum(self):
        return float(np.sum(self.window))
