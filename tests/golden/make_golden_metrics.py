"""Golden vectors of the reference's Metrics (mineral/common/metrics.py) with regex-selected info keys.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_metrics.py [/root/reference]

The reference class is driven on the CPU with hash-derived per-env values (synth.py, rebuilt by the tests): `succ` and
`dist` are one float per env.  Recorded: the windows of every tracker and result() after the last step."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402

import synth  # noqa: E402
from mineral.common.metrics import Metrics  # noqa: E402

CASES = synth.METRICS_CASES

out = {}
for name, (N, L, steps, p_done) in CASES.items():
    scores = {'rewards': torch.zeros(N), 'lengths': torch.zeros(N, dtype=torch.int64)}
    m = Metrics(scores, L, info_keys=synth.METRICS_INFO_KEYS)
    for t in range(steps):
        obs, r, d = synth.metrics_step(name, t)
        idx = torch.where(d)[0].tolist()
        m.update(0, None, obs, r, idx, {})
    res = m.result(prefix='train')
    for k, v in res.items():
        out[f'{name}/result/{k}'] = np.float64(v)
    for k, tr in m.episode_trackers.items():
        out[f'{name}/window/{k}'] = np.asarray(list(tr.window), dtype=np.float64)
    out[f'{name}/num_episodes'] = np.int64(m.num_episodes)
np.savez_compressed(os.path.join(HERE, 'g17_metrics_info_keys.npz'), **out)
print('g17_metrics_info_keys:', len(out), 'arrays')
