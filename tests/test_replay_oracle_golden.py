"""The replay oracle against the vectors recorded from the reference's buffers (bit-exact: copies and fp32 sums)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import synth  # noqa: E402
from oracle import replay_oracle as R  # noqa: E402

G = np.load(os.path.join(ROOT, 'tests', 'golden', 'g11_replay.npz'))
NAMES = ('obs', 'action', 'reward', 'next_obs', 'done')


@pytest.mark.parametrize('case', sorted(synth.REPLAY_NSTEP_CASES))
def test_nstep_matches_reference(case):
    N, O, A, nstep, gamma, horizons, _ = synth.REPLAY_NSTEP_CASES[case]
    buf = R.NStepOracle(N, O, A, nstep, gamma)
    for call in range(len(horizons)):
        inp = [t.numpy() for t in synth.replay_call_inputs(case, call, 11)]
        out = buf.add_to_buffer(*inp)
        for nm, v in zip(NAMES, out):
            ref = G[f'nstep_{case}_c{call}_{nm}']
            assert v.shape == ref.shape, (case, call, nm, v.shape, ref.shape)
            assert np.array_equal(v, ref), (case, call, nm)


@pytest.mark.parametrize('case', sorted(synth.REPLAY_RING_CASES))
def test_ring_matches_reference(case):
    cap, O, A, rows = synth.REPLAY_RING_CASES[case]
    mem = R.RingOracle(O, A, cap)
    for add in range(len(rows)):
        mem.add_to_buffer(*[t.numpy() for t in synth.replay_rows(case, add, 12)])
        state = G[f'ring_{case}_a{add}_state']
        assert (mem.next_p, mem.cur_capacity, int(mem.if_full)) == tuple(int(x) for x in state)
        k = mem.cur_capacity
        for nm, arr in zip(NAMES, (mem.obs, mem.action, mem.reward, mem.next_obs, mem.done)):
            assert np.array_equal(arr[:k], G[f'ring_{case}_a{add}_{nm}']), (case, add, nm)
    idx = G[f'ring_{case}_sample_idx']
    for nm, v in zip(NAMES, mem.gather(idx)):
        assert np.array_equal(v, G[f'ring_{case}_sample_{nm}']), (case, nm)


def test_nstep_without_an_emitted_step_fails_like_the_reference():
    buf = R.NStepOracle(2, 4, 2, nstep=5)
    z = np.zeros
    with pytest.raises(ValueError):
        buf.add_to_buffer(z((2, 3, 4), np.float32), z((2, 3, 2), np.float32), z((2, 3, 1), np.float32), z((2, 3, 4), np.float32),
                          z((2, 3, 1), np.float32))
