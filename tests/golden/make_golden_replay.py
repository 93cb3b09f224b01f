"""Golden vectors of the replay path, produced by running the REFERENCE buffers (read-only checkout) on CPU:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_replay.py [/root/reference]

`mineral.buffers` imports without stubs.  Inputs come from synth.py; the fixture holds outputs only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)

import importlib.util  # noqa: E402

import torch  # noqa: E402

import synth  # noqa: E402


def load_ref(mod, path):
    spec = importlib.util.spec_from_file_location(mod, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[mod] = m
    spec.loader.exec_module(m)
    return m


# load the two buffer modules directly (the package __init__ of the reference pulls in every agent)
pkg = type(sys)('refbuffers')
pkg.__path__ = [os.path.join(REF, 'mineral', 'buffers')]
sys.modules['refbuffers'] = pkg
simple = load_ref('refbuffers.simple_replay', os.path.join(REF, 'mineral', 'buffers', 'simple_replay.py'))
nstep_mod = load_ref('refbuffers.nstep_replay', os.path.join(REF, 'mineral', 'buffers', 'nstep_replay.py'))

out = {}
for case, (N, O, A, nstep, gamma, horizons, pdone) in synth.REPLAY_NSTEP_CASES.items():
    buf = nstep_mod.NStepReplay({'obs': (O,)}, A, N, nstep, gamma, device='cpu')
    for call in range(len(horizons)):
        obs, act, rew, nxt, done = synth.replay_call_inputs(case, call, 11)
        r = buf.add_to_buffer({'obs': obs}, act, rew, {'obs': nxt}, done)
        names = ('obs', 'action', 'reward', 'next_obs', 'done')
        vals = (r[0]['obs'], r[1], r[2], r[3]['obs'], r[4])
        for nm, v in zip(names, vals):
            out[f'nstep_{case}_c{call}_{nm}'] = v.float().numpy()

for case, (cap, O, A, rows) in synth.REPLAY_RING_CASES.items():
    mem = simple.ReplayBuffer({'obs': (O,)}, A, cap, device='cpu')
    for add in range(len(rows)):
        obs, act, rew, nxt, done = synth.replay_rows(case, add, 12)
        mem.add_to_buffer(({'obs': obs}, act, rew, {'obs': nxt}, done))
        k = mem.cur_capacity
        out[f'ring_{case}_a{add}_state'] = np.array([mem.next_p, mem.cur_capacity, int(mem.if_full)], dtype=np.int64)
        out[f'ring_{case}_a{add}_obs'] = mem.buf_obs['obs'][:k].numpy().copy()
        out[f'ring_{case}_a{add}_action'] = mem.buf_action[:k].numpy().copy()
        out[f'ring_{case}_a{add}_reward'] = mem.buf_reward[:k].numpy().copy()
        out[f'ring_{case}_a{add}_next_obs'] = mem.buf_next_obs['obs'][:k].numpy().copy()
        out[f'ring_{case}_a{add}_done'] = mem.buf_done[:k].numpy().astype(np.uint8).copy()
    # sample_batch with injected indices (torch.randint streams differ per device, so the index draw is not part of parity)
    idx = torch.from_numpy((synth.lcg((33,), 99, 1.0, np.float64) + 0.5) * mem.cur_capacity).long().clamp_(0, mem.cur_capacity - 1)
    orig = torch.randint
    torch.randint = lambda *a, **k: idx
    try:
        o, a_, r_, n_, d_ = mem.sample_batch(33, device='cpu')
    finally:
        torch.randint = orig
    out[f'ring_{case}_sample_idx'] = idx.numpy()
    for nm, v in zip(('obs', 'action', 'reward', 'next_obs', 'done'), (o['obs'], a_, r_, n_['obs'], d_)):
        out[f'ring_{case}_sample_{nm}'] = v.numpy().copy()

path = os.path.join(HERE, 'g11_replay.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')
