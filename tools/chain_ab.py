"""A/B of the small-batch kernel family (MRL_CHAIN bit mask, read once per process): gradients of one minibatch for several
networks dumped to a file, and the time of an optimizer step at the shipped minibatch of 2,048 rows.

    MRL_CHAIN=0 python tools/chain_ab.py dump /tmp/a.pt ; MRL_CHAIN=3 python tools/chain_ab.py dump /tmp/b.pt
    python tools/chain_ab.py cmp /tmp/a.pt /tmp/b.pt
    python tools/chain_ab.py time            # us per optimizer step (merge + grad + Adam), 2,048 rows, DFlexAntPPO network
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import torch  # noqa: E402

DEV = 'cuda:0'
CASES = [('mid', 64, 8, 128, 'dreamerv3_normal'), ('full', 64, 8, 256, 'dreamerv3_normal'), ('wide2', 37, 8, 148, 'dreamerv3_normal'),
         ('full', 64, 32, 2048, 'dreamerv3_normal'), ('cartpole', 37, 8, 296, 'normal'), ('elu_head', 64, 8, 256, 'dreamerv3_normal'),
         ('full_lnelu', 64, 8, 256, 'normal'), ('ln_fixed', 64, 8, 256, 'normal'), ('obs128', 100, 8, 800, 'normal')]

# AB_BIG=1: batches above 8,192 rows instead (the per-layer kernel family; ragged row counts included)
if os.environ.get('AB_BIG'):
    CASES = [('full', 1024, 32, 16384, 'dreamerv3_normal'), ('full', 2048, 32, 32768, 'dreamerv3_normal'), ('full_lnelu', 1024, 32, 16384, 'normal'),
             ('full', 1024, 32, 12000, 'dreamerv3_normal'), ('full', 1024, 32, 9999, 'dreamerv3_normal'), ('mid', 1024, 32, 16384, 'dreamerv3_normal'), ('full_obs40', 1024, 32, 12000, 'dreamerv3_normal')]

def grads_of(case):
    import test_hip_parity as T
    from mineral_amd import _hip as hip
    net, N, Tn, mb, dist = case
    spec = T.NETS[net]
    params = T.oracle_params(spec, 9)
    model = T.make_model(spec, params, dist)
    buf = T.make_buffer(spec, N, Tn, mb)
    data = T._fill_training_buffer(spec, buf, N, Tn)
    chp = hip.PpoHp(0.2, 4.0, 0.01, 1e-4, 0, 0, 0)
    L = hip.lib()
    ws = torch.zeros(int(L.mrl_workspace_bytes(model.net, mb)), dtype=torch.uint8, device=DEV)
    grads = torch.full((model.param_count,), float('nan'), device=DEV)
    metrics = torch.zeros(hip.NMETRIC, dtype=torch.float64, device=DEV)
    mean_f = data['obs'][:mb].mean(0).to(DEV).contiguous()
    std_f = (data['obs'][:mb].std(0) + 0.1).to(DEV).contiguous()
    row0 = (N * Tn // mb - 1) * mb
    hip.check(L.mrl_ppo_grad(model.net, hip.ptr(model.flat), chp, buf.c, row0, mb, hip.ptr(mean_f), hip.ptr(std_f), -5.0, 5.0,
                             hip.ptr(grads), hip.ptr(metrics), hip.ptr(ws), ws.numel(), None), 'grad')
    torch.cuda.synchronize()
    gv = model.views_of(grads)
    return {k: v.detach().cpu().clone() for k, v in gv.items()}, metrics.cpu()


def dump(path):
    out = {}
    for case in CASES:
        print('case', case, flush=True)
        g, m = grads_of(case)
        out['/'.join(map(str, case))] = (g, m)
    torch.save(out, path)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    worst = 0.0
    for k in A:
        ga, ma = A[k]
        gb, mb = B[k]
        line = []
        for name in ga:
            scale = float(ga[name].abs().max()) + 1e-12
            err = float((ga[name] - gb[name]).abs().max()) / scale
            if not (err <= 1e9):
                err = float('inf')
            worst = max(worst, err)
            if err > 1e-5:
                line.append(f'{name}:{err:.2e}')
        merr = float((ma - mb).abs().max())
        print(f'{k}: max metric diff {merr:.2e}; tensors over 1e-5 of their scale: {" ".join(line) if line else "none"}')
    print(f'worst relative difference {worst:.3e}')
    return worst


def timing(mb=2048, N=2048, Tn=32, steps=300):
    import test_hip_parity as T
    from mineral_amd import _hip as hip
    spec = T.NETS['full']
    params = T.oracle_params(spec, 9)
    model = T.make_model(spec, params)
    buf = T.make_buffer(spec, N, Tn, mb)
    T._fill_training_buffer(spec, buf, N, Tn)
    chp = hip.PpoHp(0.2, 4.0, 0.0, 1e-4, 0, 0, 0)
    L = hip.lib()
    ws = torch.zeros(int(L.mrl_workspace_bytes(model.net, mb)), dtype=torch.uint8, device=DEV)
    grads = torch.zeros(model.param_count, device=DEV)
    metrics = torch.zeros(hip.NMETRIC, dtype=torch.float64, device=DEV)
    mean_f = torch.zeros(spec['obs'], device=DEV)
    std_f = torch.ones(spec['obs'], device=DEV)
    m1, m2 = torch.zeros_like(model.flat), torch.zeros_like(model.flat)
    gn = torch.zeros(1, device=DEV)
    scratch = torch.empty(4096, dtype=torch.uint8, device=DEV)
    n_mb = N * Tn // mb

    one_call = os.environ.get('MRL_ONE_CALL', '1') != '0'

    def run(n, adam=True):
        for i in range(n):
            if adam and one_call:
                hip.check(L.mrl_ppo_step(model.net, hip.ptr(model.flat), chp, buf.c, (i % n_mb) * mb, mb, None, None, None, 0.0, hip.ptr(mean_f),
                                         hip.ptr(std_f), -5.0, 5.0, hip.ptr(grads), hip.ptr(metrics), hip.ptr(ws), ws.numel(), hip.ptr(m1), hip.ptr(m2),
                                         1e-5, 0.9, 0.999, 1e-5, 0.01, 1, i + 1, 0.5, hip.ptr(gn), hip.ptr(scratch), None), 'step')
                continue
            hip.check(L.mrl_ppo_grad(model.net, hip.ptr(model.flat), chp, buf.c, (i % n_mb) * mb, mb, hip.ptr(mean_f), hip.ptr(std_f), -5.0, 5.0,
                                     hip.ptr(grads), hip.ptr(metrics), hip.ptr(ws), ws.numel(), None), 'grad')
            if adam:
                hip.check(L.mrl_adam_step(hip.ptr(model.flat), hip.ptr(grads), hip.ptr(m1), hip.ptr(m2), model.param_count, 1e-5, 0.9, 0.999,
                                          1e-5, 0.01, 1, i + 1, 0.5, 1.0, hip.ptr(gn), hip.ptr(scratch), None), 'adam')
    run(20)
    torch.cuda.synchronize()
    for adam in (True, False):
        t0 = time.perf_counter()
        run(steps, adam)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'MRL_CHAIN={os.environ.get("MRL_CHAIN", "default")} mb={mb}: {1e6 * dt / steps:.1f} us per step ({("one call" if one_call else "grad + Adam") if adam else "grad only"})', flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'dump':
        dump(sys.argv[2])
    elif sys.argv[1] == 'cmp':
        sys.exit(0 if cmp(sys.argv[2], sys.argv[3]) < 1e-3 else 1)
    else:
        timing(*(int(a) for a in sys.argv[2:]))
