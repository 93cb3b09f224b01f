"""The SAC oracle against the run recorded from the reference agent (tests/golden/g12_sac_*.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import sac_harness as H  # noqa: E402
import synth  # noqa: E402


@pytest.mark.parametrize('case', sorted(synth.SAC_CASES))
def test_sac_training_run_matches_reference(case):
    G = np.load(os.path.join(ROOT, 'tests', 'golden', f'g12_sac_{case}.npz'))
    epochs = synth.SAC_CASES[case][-1]
    seen = []

    def check(e, st):
        seen.append(e)
        assert [st['counters'][k] for k in ('rand', 'noise', 'idx')] == [int(x) for x in G[f'e{e}_counters']]
        # replay contents: copies and fp32 n-step sums in the reference's order.  The rows of the warm-up (uniform actions) do
        # not depend on the policy and must match bit for bit; later rows carry policy actions, whose last bits depend on
        # how the CPU GEMMs were threaded when the vectors were recorded
        mem = st['mem']
        assert (mem.next_p, mem.cur_capacity, int(mem.if_full)) == tuple(int(x) for x in G[f'e{e}_memory_state'])
        k = mem.cur_capacity
        assert np.array_equal(mem.done[:k], G[f'e{e}_memory_done'])
        assert np.allclose(mem.reward[:k], G[f'e{e}_memory_reward'], rtol=0, atol=1e-6)
        if e == 0:
            N, hpc = synth.SAC_CASES[case][0], synth.SAC_CASES[case][5]
            w = N * (hpc['warm_up'] - hpc['nstep'] + 1)
            for got, key in ((mem.obs, 'obs'), (mem.action, 'action'), (mem.next_obs, 'next_obs'), (mem.reward, 'reward')):
                assert np.array_equal(got[:w], G[f'e0_memory_{key}'][:w]), key
                assert np.allclose(got[:k], G[f'e0_memory_{key}'], rtol=0, atol=1e-5), key
        # running statistics, parameters, log_alpha: same torch ops in the same order; tolerance for op fusion only
        for key in ('running_mean', 'running_var', 'running_count'):
            assert torch.allclose(st['obs_rms'][key], torch.from_numpy(G[f'e{e}_obs_rms.{key}']), rtol=1e-6, atol=1e-7), key
        for name, v in st['pa'].items():
            assert torch.allclose(v, torch.from_numpy(G[f'e{e}_actor.{name}']), rtol=2e-5, atol=2e-6), (e, name)
        for name, v in st['pc'].items():
            assert torch.allclose(v, torch.from_numpy(G[f'e{e}_critic.{name}']), rtol=2e-5, atol=2e-6), (e, name)
        assert abs(float(st['log_alpha']) - float(G[f'e{e}_log_alpha'])) < 1e-6
        for key in ('loss/critic', 'loss/actor', 'loss/alpha', 'entropy', 'grad_norm/critic', 'grad_norm/actor'):
            got = torch.stack([r[key] for r in st['rec']]).float().numpy()
            ref = G[f'e{e}_{key.replace("/", "_")}']
            assert np.allclose(got, ref, rtol=2e-4, atol=1e-5), (e, key, got, ref)
        if e == epochs - 1:
            for name, v in st['pt'].items():
                assert torch.allclose(v, torch.from_numpy(G[f'e{e}_critic_target.{name}']), rtol=2e-5, atol=2e-6), name

    H.run_oracle(case, check)
    assert seen == list(range(epochs))
