"""HBM rate of the replay kernels at the SAC configuration of the baseline (4096 envs x 32 steps per add, nstep 3,
1M-transition ring, obs 64, act 12), with the CPU oracle timed beside it on a bounded sample.

    python tools/replay_bench.py [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mineral_amd.buffers import NStepReplay, ReplayBuffer  # noqa: E402

DEV = 'cuda:0'
N, T, O, A, NSTEP, CAP = 4096, 32, 64, 12, 3, 1_000_000
ROW = (2 * O + A + 2) * 4            # bytes of one stored transition (obs, next_obs, action, reward, done as float)
PEAK = 8000.0                        # GB/s, MI355X HBM3E


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    obs, nxt = torch.randn(N, T, O, device=DEV, generator=g), torch.randn(N, T, O, device=DEV, generator=g)
    act, rew = torch.rand(N, T, A, device=DEV, generator=g) * 2 - 1, torch.randn(N, T, 1, device=DEV, generator=g)
    done = (torch.rand(N, T, 1, device=DEV, generator=g) < 0.001).float()
    nb = NStepReplay({'obs': (O,)}, A, N, NSTEP, 0.99, device=DEV)
    mem = ReplayBuffer({'obs': (O,)}, A, CAP, device=DEV)
    traj = nb.add_to_buffer({'obs': obs}, act, rew, {'obs': nxt}, done)
    rows = N * T
    res = {'config': f'{N} envs x {T} steps per add, nstep {NSTEP}, ring of {CAP} transitions, obs {O}, act {A}', 'peak_GBps': PEAK, 'kernels': {}}

    t = timed(lambda: nb.add_to_buffer({'obs': obs}, act, rew, {'obs': nxt}, done), 20)
    b = rows * (ROW + (2 * O + A) * 4 + NSTEP * 8)        # write one row, read obs + action + next_obs + nstep rewards/dones
    res['kernels']['nstep_emit (+ history)'] = {'us': t * 1e6, 'algorithmic_bytes': b, 'GBps': b / t / 1e9, 'frac_of_peak': b / t / 1e9 / PEAK}
    for _ in range(8):
        mem.add_to_buffer(traj)                           # fill the ring
    t = timed(lambda: mem.add_to_buffer(traj), 20)
    b = traj[2].shape[0] * (2 * ROW - 3)
    res['kernels']['replay_write'] = {'us': t * 1e6, 'algorithmic_bytes': b, 'GBps': b / t / 1e9, 'frac_of_peak': b / t / 1e9 / PEAK}
    for batch in (2048, 65536, 1 << 20):
        idx = torch.randint(CAP, (batch,), device=DEV, generator=g)
        t = timed(lambda: mem.sample_batch(batch, device=DEV, indices=idx), 20)
        b = batch * (2 * ROW - 3 + 8)
        res['kernels'][f'replay_gather batch {batch}'] = {'us': t * 1e6, 'algorithmic_bytes': b, 'GBps': b / t / 1e9, 'frac_of_peak': b / t / 1e9 / PEAK}

    # CPU leg: the oracle (numpy restatement of the reference buffers) on a bounded sample of the same workload
    from oracle import replay_oracle as R
    n_cpu = 256
    o_, a_, r_, x_, d_ = (t_[:n_cpu].cpu().numpy() for t_ in (obs, act, rew, nxt, done))
    ora = R.NStepOracle(n_cpu, O, A, NSTEP, 0.99)
    ring = R.RingOracle(O, A, 100_000)
    t0 = time.perf_counter()
    out = ora.add_to_buffer(o_, a_, r_, x_, d_)
    ring.add_to_buffer(*out)
    ring.gather(np.random.RandomState(0).randint(0, ring.cur_capacity, 2048))
    dt = time.perf_counter() - t0
    res['cpu_baseline'] = {'value': n_cpu * T / dt, 'unit': 'transitions/s through n-step + ring write (+ one 2048 gather)', 'cores': 1, 'kind': 'port',
                           'sample': f'{n_cpu} envs x {T} steps, {dt:.2f} s'}
    emit_us = res['kernels']['nstep_emit (+ history)']['us'] + res['kernels']['replay_write']['us']
    res['gpu_transitions_per_s'] = rows / (emit_us * 1e-6)
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
