"""Launcher helpers with the behaviour of mineral/scripts/utils.py:17-93 (seeding, thread limits, numpy printing)."""

import os
import random

import numpy as np


def set_np_formatting():
    np.set_printoptions(edgeitems=30, infstr='inf', linewidth=4000, nanstr='nan', precision=2, suppress=False, threshold=10000,
                        formatter=None)


def set_seed(seed, torch_deterministic=False, rank=0):
    """Seed python / numpy / torch with seed + rank (utils.py:43-78); -1 draws a seed."""
    import torch
    if seed == -1 and torch_deterministic:
        seed = 42 + rank
    elif seed == -1:
        seed = np.random.randint(0, 10000)
    else:
        seed = seed + rank
    print(f'Setting seed: {seed}')
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if torch_deterministic:
        torch.use_deterministic_algorithms(True)
    return seed


def limit_threads(n=1):
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'VECLIB_MAXIMUM_THREADS', 'NUMEXPR_NUM_THREADS'):
        os.environ[var] = str(n)


def launcher_seeds(seed):
    """Sub-seeds drawn deterministically from config.seed (run.py:122-127): seed 42 -> setup 121958, train 671155,
    eval 131932, env_train 365838, env_eval 259178."""
    names = ['setup', 'train', 'eval', 'env_train', 'env_eval']
    draws = np.random.RandomState(seed).randint(0, int(1e6), len(names))
    return {k: int(draws[i]) for i, k in enumerate(names)}
