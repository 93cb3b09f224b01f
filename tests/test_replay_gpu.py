"""HIP replay path (n-step FIFO, ring write, batch gather) through the product classes / the C ABI: bit-exact against the
vectors recorded from the reference's buffers, against the oracle on random inputs, and size-independent properties at
the 1M-transition / 4096-env configuration."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import synth  # noqa: E402
from oracle import replay_oracle as R  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
NAMES = ('obs', 'action', 'reward', 'next_obs', 'done')


@pytest.fixture(scope='module')
def G():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'g11_replay.npz'))


def unpack(ret):
    o, a, r, n, d = ret
    return (o['obs'], a, r, n['obs'], d)


@pytest.mark.parametrize('case', sorted(synth.REPLAY_NSTEP_CASES))
def test_nstep_bit_exact_vs_reference(case, G):
    from mineral_amd.buffers import NStepReplay
    N, O, A, nstep, gamma, horizons, _ = synth.REPLAY_NSTEP_CASES[case]
    buf = NStepReplay({'obs': (O,)}, A, N, nstep, gamma, device=DEV)
    for call in range(len(horizons)):
        obs, act, rew, nxt, done = (t.to(DEV) for t in synth.replay_call_inputs(case, call, 11))
        out = unpack(buf.add_to_buffer({'obs': obs}, act, rew, {'obs': nxt}, done))
        for nm, v in zip(NAMES, out):
            ref = G[f'nstep_{case}_c{call}_{nm}']
            assert tuple(v.shape) == ref.shape, (case, call, nm)
            assert np.array_equal(v.float().cpu().numpy(), ref), (case, call, nm)


@pytest.mark.parametrize('case', sorted(synth.REPLAY_RING_CASES))
def test_ring_bit_exact_vs_reference(case, G):
    from mineral_amd.buffers import ReplayBuffer
    cap, O, A, rows = synth.REPLAY_RING_CASES[case]
    mem = ReplayBuffer({'obs': (O,)}, A, cap, device=DEV)
    for add in range(len(rows)):
        obs, act, rew, nxt, done = (t.to(DEV) for t in synth.replay_rows(case, add, 12))
        mem.add_to_buffer(({'obs': obs}, act, rew, {'obs': nxt}, done))
        state = G[f'ring_{case}_a{add}_state']
        assert (mem.next_p, mem.cur_capacity, int(mem.if_full)) == tuple(int(x) for x in state)
        k = mem.cur_capacity
        for nm, t in zip(NAMES, (mem.buf_obs['obs'], mem.buf_action, mem.buf_reward, mem.buf_next_obs['obs'], mem.buf_done)):
            assert np.array_equal(t[:k].cpu().numpy(), G[f'ring_{case}_a{add}_{nm}']), (case, add, nm)
    idx = torch.from_numpy(G[f'ring_{case}_sample_idx']).to(DEV)
    out = unpack(mem.sample_batch(len(idx), device=DEV, indices=idx))
    for nm, v in zip(NAMES, out):
        assert v.dtype == torch.float32
        assert np.array_equal(v.cpu().numpy(), G[f'ring_{case}_sample_{nm}']), (case, nm)


@pytest.mark.parametrize('nstep,N,O,A,horizons', [(3, 37, 12, 3, (5, 1, 1, 9)), (7, 10, 4, 16, (7, 2, 30)), (2, 129, 64, 12, (2, 1, 3))])
def test_nstep_vs_oracle_random(nstep, N, O, A, horizons):
    """Ragged sizes, horizons shorter than the window, done flags at every window position."""
    from mineral_amd.buffers import NStepReplay
    g = torch.Generator().manual_seed(nstep * 100 + N)
    buf = NStepReplay({'obs': (O,)}, A, N, nstep, 0.95, device=DEV)
    ora = R.NStepOracle(N, O, A, nstep, 0.95)
    for T in horizons:
        obs, nxt = torch.randn(N, T, O, generator=g), torch.randn(N, T, O, generator=g)
        act, rew = torch.rand(N, T, A, generator=g) * 2 - 1, torch.randn(N, T, 1, generator=g)
        done = (torch.rand(N, T, 1, generator=g) < 0.3).float()
        out = unpack(buf.add_to_buffer({'obs': obs.to(DEV)}, act.to(DEV), rew.to(DEV), {'obs': nxt.to(DEV)}, done.to(DEV)))
        ref = ora.add_to_buffer(obs.numpy(), act.numpy(), rew.numpy(), nxt.numpy(), done.numpy())
        for nm, v, r in zip(NAMES, out, ref):
            assert np.array_equal(v.cpu().numpy(), r), (nstep, T, nm)


def test_abi_rejects_bad_arguments():
    from mineral_amd import _hip
    from mineral_amd.buffers import NStepReplay, ReplayBuffer
    L = _hip.lib()
    mem = ReplayBuffer({'obs': (8,)}, 2, 16, device=DEV)
    z = torch.zeros
    with pytest.raises(ValueError):
        mem.add_to_buffer(({'obs': z(17, 8, device=DEV)}, z(17, 2, device=DEV), z(17, 1, device=DEV), {'obs': z(17, 8, device=DEV)}, z(17, 1, device=DEV)))
    assert L.mrl_replay_write(None, 16, 0, None, 1, None) == -1 and b'NULL' in L.mrl_last_error()
    with pytest.raises(NotImplementedError):
        NStepReplay({'obs': (8,)}, 2, 4, nstep=8, device=DEV)
    with pytest.raises(NotImplementedError):
        ReplayBuffer({'obs': (6,)}, 2, 16, device=DEV)
    with pytest.raises(_hip.HipLibraryError):
        ReplayBuffer({'obs': (8,)}, 2, 16, device='cpu')
    buf = NStepReplay({'obs': (8,)}, 2, 4, nstep=5, device=DEV)
    with pytest.raises(ValueError):   # fewer than nstep steps pushed: the reference fails here as well
        buf.add_to_buffer({'obs': z(4, 3, 8, device=DEV)}, z(4, 3, 2, device=DEV), z(4, 3, 1, device=DEV), {'obs': z(4, 3, 8, device=DEV)}, z(4, 3, 1, device=DEV))


def test_full_size_properties():
    """BASELINE-size run (4096 envs x 32 steps per add, nstep 3, 1M-transition ring, obs 64, act 12): the ring after
    several wraps holds exactly the rows an index-map model predicts, gathers equal direct indexing, n-step rewards equal
    the closed form, and a second pass reproduces the same bits."""
    from mineral_amd.buffers import NStepReplay, ReplayBuffer
    N, T, O, A, nstep, cap = 4096, 32, 64, 12, 3, 1_000_000
    g = torch.Generator(device=DEV).manual_seed(5)

    def run():
        gg = torch.Generator(device=DEV).manual_seed(5)
        nb = NStepReplay({'obs': (O,)}, A, N, nstep, 0.99, device=DEV)
        mem = ReplayBuffer({'obs': (O,)}, A, cap, device=DEV)
        tag = 0
        where = torch.full((cap,), -1, dtype=torch.int64, device=DEV)   # which emitted row sits in each slot
        pos = 0
        last = None
        for call in range(9):   # 9 x ~131k rows > 1M: wraps once
            obs = torch.randn(N, T, O, device=DEV, generator=gg)
            nxt = torch.randn(N, T, O, device=DEV, generator=gg)
            act = torch.rand(N, T, A, device=DEV, generator=gg) * 2 - 1
            rew = torch.randn(N, T, 1, device=DEV, generator=gg)
            done = (torch.rand(N, T, 1, device=DEV, generator=gg) < 0.001).float()
            o, a, r, n_, d = nb.add_to_buffer({'obs': obs}, act, rew, {'obs': nxt}, done)
            rows = r.shape[0]
            assert rows == (T - (nstep - 1 if call == 0 else 0)) * N
            if call == 3:   # closed form of the n-step return where the window holds no done
                first = 0
                e = torch.arange(T, device=DEV)
                # emitted step i of this call uses call steps i-2, i-1, i (i >= 2 lie fully inside the call)
                rr = rew[:, :, 0]
                dd = done[:, :, 0]
                for i in (2, 17, 31):
                    w = rr[:, i - 2:i + 1]
                    clean = dd[:, i - 2:i + 1].sum(1) == 0
                    ga = nb.gamma_array.view(-1)
                    expect = (w[:, 0] * ga[0] + w[:, 1] * ga[1]) + w[:, 2] * ga[2]
                    got = r.view(T, N)[i][:, None][:, 0] if False else r.view(-1, N)[i]
                    assert torch.equal(got[clean], expect[clean])
                    assert torch.equal(o['obs'].view(-1, N, O)[i], obs[:, i - 2])
                    assert torch.equal(a.view(-1, N, A)[i], act[:, i - 2])
                    assert torch.equal(n_['obs'].view(-1, N, O)[i][clean], nxt[:, i][clean])
            mem.add_to_buffer((o, a, r, n_, d))
            slots = (pos + torch.arange(rows, device=DEV)) % cap
            where[slots] = tag + torch.arange(rows, device=DEV)
            pos = (pos + rows) % cap
            tag += rows
            last = (o['obs'], a, r, n_['obs'], d, slots)
        return mem, where, last

    mem, where, last = run()
    assert mem.if_full and mem.cur_capacity == cap and mem.next_p == (9 * T - 2) * N - cap
    # the rows of the last add sit where the index map says
    lo, la, lr, ln, ld, slots = last
    assert torch.equal(mem.buf_obs['obs'][slots], lo) and torch.equal(mem.buf_action[slots], la)
    assert torch.equal(mem.buf_reward[slots], lr) and torch.equal(mem.buf_next_obs['obs'][slots], ln)
    assert torch.equal(mem.buf_done[slots].float(), ld)
    assert int((where >= 0).sum()) == cap
    # gather == direct indexing, for a million random indices
    idx = torch.randint(cap, (1 << 20,), device=DEV, generator=g)
    o, a, r, n_, d = mem.sample_batch(idx.numel(), device=DEV, indices=idx)
    assert torch.equal(o['obs'], mem.buf_obs['obs'][idx]) and torch.equal(n_['obs'], mem.buf_next_obs['obs'][idx])
    assert torch.equal(a, mem.buf_action[idx]) and torch.equal(r, mem.buf_reward[idx]) and torch.equal(d, mem.buf_done[idx].float())
    # bit-reproducible
    mem2, _, _ = run()
    for t1, t2 in ((mem.buf_obs['obs'], mem2.buf_obs['obs']), (mem.buf_reward, mem2.buf_reward), (mem.buf_done, mem2.buf_done)):
        assert torch.equal(t1, t2)
