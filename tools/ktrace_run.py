"""Cycle stamps inside one GEMM block, from the instrumented diagnostic build (`make ktrace`).

    MRL_KTRACE='<128, 128, 16, 2, 2, 1, 1, 2, 2, 2>' MRL_KTRACE_M=512 python tools/ktrace_run.py

runs a few mrl_ppo_grad calls at the benchmark's minibatch shape with random data and prints, for block 8 of the
selected GEMM instantiation, the cycles each wave spent per reduction step before the barrier and waiting in it.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mineral_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.path.join(ROOT, 'tools', 'bin', 'libmineral_ppo_ktrace.so')

from mineral_amd.agents.ppo.experience import ExperienceBuffer  # noqa: E402
from mineral_amd.agents.ppo.models import ActorCritic  # noqa: E402


def main():
    dev = 'cuda:0'
    obs, act, units = 64, 12, [512, 256, 128]
    N, T, mb = 2048, 32, 32768
    model = ActorCritic({'obs': (obs,)}, act, mlp_kwargs=dict(units=units, norm_type='LayerNorm', act_type='SiLU'), fixed_sigma=False,
                        actor_dist_kwargs=dict(dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0), device=dev)
    buf = ExperienceBuffer(N, T, N * T, mb, {'obs': (obs,)}, act, dev)
    B = N * T
    g = torch.Generator(device=dev).manual_seed(1)
    s = buf.storage_dict
    s['obses']['obs'].view(B, -1).copy_(torch.randn(B, obs, device=dev, generator=g))
    s['actions'].view(B, act).copy_(torch.randn(B, act, device=dev, generator=g) * 0.5)
    s['mu'].view(B, act).copy_(torch.tanh(torch.randn(B, act, device=dev, generator=g) * 0.3))
    s['sigma'].view(B, act).copy_(torch.rand(B, act, device=dev, generator=g) * 0.5 + 0.3)
    buf._flat['neglogp'].copy_(torch.randn(B, device=dev, generator=g) * 0.3 + 8.0)
    for k in ('advantages', 'returns', 'values'):
        buf._flat[k].copy_(torch.randn(B, device=dev, generator=g))
    L = _hip.lib()
    chp = _hip.PpoHp(0.2, 4.0, 0.0, 1e-4, 0, 0, 0)
    ws = torch.empty(int(L.mrl_workspace_bytes(model.net, mb)), dtype=torch.uint8, device=dev)
    grads = torch.zeros(model.param_count, device=dev)
    metrics = torch.zeros(_hip.NMETRIC, dtype=torch.float64, device=dev)
    mean_f = torch.zeros(obs, device=dev)
    std_f = torch.ones(obs, device=dev)
    for start in (0, mb) * 10:
        _hip.check(L.mrl_ppo_grad(model.net, _hip.ptr(model.flat), chp, buf.c, start, mb, _hip.ptr(mean_f), _hip.ptr(std_f), -5.0, 5.0,
                                  _hip.ptr(grads), _hip.ptr(metrics), _hip.ptr(ws), ws.numel(), None), 'mrl_ppo_grad')
    torch.cuda.synchronize()
    slots = 1024
    host = (C.c_ulonglong * (8 * slots))()
    L.mrl_debug_ktrace.restype = C.c_int
    assert L.mrl_debug_ktrace(host) == 0
    t = np.frombuffer(host, dtype=np.uint64).reshape(8, slots).astype(np.int64)[:4]   # the first four waves
    nz = int((t[0, :1000] != 0).sum())
    nsteps = (nz - 1) // 2
    e = t[:, 1000:1003]
    for w in range(4):
        print(f'wave {w}: entry->main loop {int(t[w, 0] - e[w, 0])} cycles, main loop {int(e[w, 1] - t[w, 0])}, epilogue {int(e[w, 2] - e[w, 1])}, whole block {int(e[w, 2] - e[w, 0])}')
    kb = (C.c_ulonglong * (2 * 8192))()
    assert L.mrl_debug_kblocks(kb) == 0
    b = np.frombuffer(kb, dtype=np.uint64).reshape(8192, 2).astype(np.int64)
    ids = np.nonzero((b[:, 0] != 0) & (b[:, 1] != 0))[0]
    b = b[ids]
    if len(b):
        t0 = b[:, 0].min()
        st, en = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0
        q = lambda x: ' '.join(f'{v:.1f}' for v in np.percentile(x, [0, 10, 50, 90, 100]))
        print('   median duration by blockIdx%8 (XCD):', ' '.join(f'{np.median((en - st)[ids % 8 == x]):.1f}' for x in range(8)))
        d = en - st
        if os.environ.get('MRL_KTRACE_DUMP'):
            np.save(os.environ['MRL_KTRACE_DUMP'], np.stack([ids, st, en], 1))
        order = np.argsort(d)
        print('   slowest block ids:', ids[order[-16:]].tolist())
        print('   fastest block ids:', ids[order[:16]].tolist())
        print(f'{len(b)} blocks (us, percentiles 0/10/50/90/100): start {q(st)} | end {q(en)} | duration {q(en - st)}')
    if t[0, 1003]:
        x = t[:, 1001:1007]
        for w in range(4):
            print(f'wave {w} LayerNorm epilogue: sums {int(x[w, 2] - x[w, 0])}, barrier {int(x[w, 3] - x[w, 2])}, deviations {int(x[w, 4] - x[w, 3])}, barrier {int(x[w, 5] - x[w, 4])}, rstd + stores {int(t[w, 1002] - x[w, 5])}')
    print(f'kernel {os.environ.get("MRL_KTRACE")} M={os.environ.get("MRL_KTRACE_M")}: {nsteps} reduction steps traced')
    if nsteps < 2:
        return
    for w in range(4):
        tw = t[w]
        arrive = tw[1:1 + 2 * nsteps:2]    # stamp before the barrier of step st
        leave = tw[2:2 + 2 * nsteps:2]     # stamp after it
        work = np.empty(nsteps, dtype=np.int64)
        work[0] = arrive[0] - tw[0]
        work[1:] = arrive[1:] - leave[:-1]
        wait = leave - arrive
        print(f'wave {w}: total {int(leave[-1] - tw[0])} cycles; per step: work median {int(np.median(work))} (min {int(work.min())}, max {int(work.max())}), '
              f'barrier wait median {int(np.median(wait))} (max {int(wait.max())})')
        if w == 0:
            print('   work per step:', ' '.join(str(int(x)) for x in work))
            top = tw[400:400 + nsteps]
            print('   first sub-step (top -> barrier):', ' '.join(str(int(x)) for x in (arrive - top)))
            print('   last sub-step (barrier -> next top):', ' '.join(str(int(x)) for x in (top[1:] - leave[:-1])))
    skew = t[:, 1:1 + 2 * nsteps:2]
    print('arrival skew between the four waves of the block (max - min), median over steps:', int(np.median(skew.max(0) - skew.min(0))))


if __name__ == '__main__':
    main()
