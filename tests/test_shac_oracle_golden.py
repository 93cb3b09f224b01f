"""The SHAC critic-side oracle against vectors recorded from the reference's own methods (tests/golden/g13_shac_critic.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import synth  # noqa: E402
from oracle import ppo_oracle as P  # noqa: E402
from oracle import shac_oracle as S  # noqa: E402

G = np.load(os.path.join(ROOT, 'tests', 'golden', 'g13_shac_critic.npz'))


@pytest.mark.parametrize('case', sorted(synth.SHAC_CASES))
def test_target_values_bit_exact(case):
    obs, rew, done, nv = synth.shac_inputs(case)
    for method in ('td-lambda', 'one-step'):
        got = S.target_values(rew, done, nv, 0.99, 0.95, method)
        assert np.array_equal(got.numpy(), G[f'{case}_targets_{method}']), method


@pytest.mark.parametrize('case', sorted(synth.SHAC_CASES))
def test_update_critic_and_soft_update(case):
    N, T, O, units, n_critics, n_batches, iters, optim_type, lr, max_norm = synth.SHAC_CASES[case]
    obs, rew, done, nv = synth.shac_inputs(case)
    targets = torch.from_numpy(G[f'{case}_targets_td-lambda'])
    bs = N * T // n_batches
    batches = S.dataset_batches(obs, targets, bs)
    assert [len(batches), bs] == [int(x) for x in G[f'{case}_dataset_len']]
    assert np.array_equal(batches[-1][1].numpy(), G[f'{case}_last_batch_targets'])      # time-major flatten, short last batch
    pc = synth.make_params(S.critic_shapes(O, units, n_critics), 61)
    st = P.adam_state_init(pc)
    res = S.update_critic(pc, st, batches, units, n_critics, iters, lr, optim_type, (0.7, 0.95), max_norm)
    for k, v in res.items():
        assert np.allclose(torch.stack(v).numpy(), G[f'{case}_{k.replace("/", "_")}'], rtol=2e-5, atol=1e-6), k
    for k, v in pc.items():
        assert torch.allclose(v, torch.from_numpy(G[f'{case}_critic.{k}']), rtol=2e-5, atol=2e-6), k
    pt = synth.make_params(S.critic_shapes(O, units, n_critics), 62)
    S.soft_update(pc, pt, 0.2)
    for k, v in pt.items():
        assert torch.allclose(v, torch.from_numpy(G[f'{case}_target.{k}']), rtol=2e-5, atol=2e-6), k



@pytest.mark.parametrize('case', sorted(synth.SHAC_CASES))
def test_entropy_target_values_bit_exact(case):
    """compute_target_values_with_entropy (shac.py:893-916) for both methods and the four entropy-transform combinations."""
    GE = np.load(os.path.join(ROOT, 'tests', 'golden', 'g14_shac_entropy_targets.npz'))
    N, T = synth.SHAC_CASES[case][:2]
    obs, rew, done, nv = synth.shac_inputs(case)
    logp = synth.gaussianish((T, N), 23) * 1.5 - 2.0
    for method in ('td-lambda', 'one-step'):
        for offset in (False, True):
            for scale in (False, True):
                got = S.target_values_with_entropy(rew, done, nv, -1.0 * logp, 0.05, 0.99, 0.95, method, offset, scale, -3.0)
                assert np.array_equal(got.numpy(), GE[f'{case}_{method}_o{int(offset)}s{int(scale)}']), (method, offset, scale)


@pytest.mark.parametrize('case', sorted(synth.SHAC_V1_CASES))
def test_update_critic_single_layernorm_elu(case):
    """The single `Critic` of SHAC v1 (models.py:149-178, LayerNorm + ELU) through update_critic, against the reference's run."""
    G1 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g15_shac_v1_critic.npz'))
    N, T, O, units, _one, n_batches, iters, optim_type, lr, max_norm = synth.SHAC_V1_CASES[case]
    obs, rew, done, nv = synth.shac_inputs(case)
    targets = S.target_values(rew, done, nv, 0.99, 0.95)
    pc = synth.make_params(S.critic_shapes(O, units, 1, single=True), 63)
    v0 = S.critic_values(pc, obs.reshape(-1, O), units, 1, 'ELU', single=True)[0]
    assert np.allclose(v0.numpy(), G1[f'{case}_values0'], rtol=1e-5, atol=1e-6)
    st = P.adam_state_init(pc)
    res = S.update_critic(pc, st, S.dataset_batches(obs, targets, N * T // n_batches), units, 1, iters, lr, optim_type, (0.7, 0.95), max_norm,
                          act_type='ELU', single=True)
    for k, v in res.items():
        assert np.allclose(torch.stack(v).numpy(), G1[f'{case}_{k.replace("/", "_")}'], rtol=2e-5, atol=1e-6), k
    for k, v in pc.items():
        assert torch.allclose(v, torch.from_numpy(G1[f'{case}_critic.{k}']), rtol=2e-5, atol=2e-6), k
