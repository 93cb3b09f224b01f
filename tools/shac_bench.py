"""Critic side of SHAC on the device: TD(lambda) target scan and `update_critic` (critic_iterations x num_critic_batches
optimizer steps), at the reference's DFlexAntSHAC2 shape (64 envs), at 4096 envs, and with the single LayerNorm+ELU critic of
DFlexAntSHAC.yaml at 8192 envs; the CPU oracle is timed beside it.

    python tools/shac_bench.py [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import torch  # noqa: E402

T, O, UNITS, NC, NB, ITERS = 32, 64, [64, 64], 2, 4, 16     # cfgs/agent/DFlexAntSHAC2.yaml:22-42 (obs padded to 64)
DEV = 'cuda:0'


def device_case(N, v1=False):
    from mineral_amd.agents.diffrl import Critic, CriticDataset, CriticLearner, EnsembleCritic, compute_target_values
    g = torch.Generator(device=DEV).manual_seed(0)
    obs = torch.randn(T, N, O, device=DEV, generator=g)
    rew = torch.randn(T, N, device=DEV, generator=g)
    nv = torch.randn(T, N, device=DEV, generator=g)
    done = (torch.rand(T, N, device=DEV, generator=g) < 0.02).float()
    if v1:   # cfgs/agent/DFlexAntSHAC.yaml:19-52
        critic = Critic(O, None, mlp_kwargs=dict(units=UNITS, norm_type='LayerNorm', act_type='ELU'), device=DEV)
        learner = CriticLearner(critic, 'Adam', dict(lr=2e-3, betas=(0.7, 0.95)), critic_iterations=ITERS, max_grad_norm=1.0, target_critic_alpha=0.2)
    else:
        critic = EnsembleCritic(O, None, n_critics=NC, mlp_kwargs=dict(units=UNITS, norm_type='LayerNorm', act_type='SiLU'), weight_init='orthogonalg1',
                                device=DEV)
        learner = CriticLearner(critic, 'AdamW', dict(lr=5e-4, betas=(0.7, 0.95)), critic_iterations=ITERS, max_grad_norm=0.5, no_target_critic=True)
    out = torch.empty_like(rew)
    for _ in range(3):
        compute_target_values(rew, done, nv, 0.99, 0.95, 'td-lambda', out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        compute_target_values(rew, done, nv, 0.99, 0.95, 'td-lambda', out=out)
    torch.cuda.synchronize()
    t_scan = (time.perf_counter() - t0) / reps
    ds = CriticDataset(N * T // NB, {'obs': obs}, out, drop_last=False)
    learner.update_critic(ds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        learner.update_critic(ds)
        learner.update_target()
    torch.cuda.synchronize()
    t_upd = (time.perf_counter() - t0) / 3
    return {'num_envs': N, 'critic': 'Critic LayerNorm+ELU' if v1 else 'EnsembleCritic x2 LayerNorm+SiLU', 'samples': N * T, 'tdlambda_us': t_scan * 1e6, 'tdlambda_GBps': 16.0 * N * T / t_scan / 1e9,
            'update_critic_ms': t_upd * 1e3, 'optimizer_steps': ITERS * NB, 'critic_samples_per_s': N * T * ITERS / t_upd}


def scan_case(N):
    """TD(lambda) scan alone at a size where HBM, not the launch, sets the time (16 bytes per (t, env))."""
    from mineral_amd.agents.diffrl import compute_target_values
    g = torch.Generator(device=DEV).manual_seed(0)
    rew = torch.randn(T, N, device=DEV, generator=g)
    nv = torch.randn(T, N, device=DEV, generator=g)
    done = (torch.rand(T, N, device=DEV, generator=g) < 0.02).float()
    out = torch.empty_like(rew)
    for _ in range(3):
        compute_target_values(rew, done, nv, 0.99, 0.95, 'td-lambda', out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        compute_target_values(rew, done, nv, 0.99, 0.95, 'td-lambda', out=out)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 20
    return {'num_envs': N, 'tdlambda_us': t * 1e6, 'tdlambda_GBps': 16.0 * N * T / t / 1e9, 'bound': 'hbm', 'peak_GBps': 8000}


def cpu_case(N, threads, v1=False):
    import synth
    from oracle import ppo_oracle as P
    from oracle import shac_oracle as S
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    obs, rew, nv = torch.randn(T, N, O, generator=g), torch.randn(T, N, generator=g), torch.randn(T, N, generator=g)
    done = (torch.rand(T, N, generator=g) < 0.02).float()
    t0 = time.perf_counter()
    tgt = S.target_values(rew, done, nv, 0.99, 0.95)
    t_scan = time.perf_counter() - t0
    pc = synth.make_params(S.critic_shapes(O, UNITS, 1 if v1 else NC, single=v1), 5)
    st = P.adam_state_init(pc)
    batches = S.dataset_batches(obs, tgt, N * T // NB)
    t0 = time.perf_counter()
    if v1:
        S.update_critic(pc, st, batches, UNITS, 1, ITERS, 2e-3, 'Adam', (0.7, 0.95), 1.0, act_type='ELU', single=True)
    else:
        S.update_critic(pc, st, batches, UNITS, NC, ITERS, 5e-4, 'AdamW', (0.7, 0.95), 0.5)
    t_upd = time.perf_counter() - t0
    return {'num_envs': N, 'critic': 'Critic LayerNorm+ELU' if v1 else 'EnsembleCritic x2 LayerNorm+SiLU', 'cores': threads, 'kind': 'port', 'tdlambda_us': t_scan * 1e6, 'update_critic_ms': t_upd * 1e3,
            'critic_samples_per_s': N * T * ITERS / t_upd}


def main():
    cores = min(16, os.cpu_count() or 1)
    res = {'metric': 'SHAC critic side: update_critic samples/s and TD(lambda) scan; critics [64,64]',
           'device': [device_case(64), device_case(4096), device_case(8192, v1=True)], 'scan': scan_case(1 << 20),
           'cpu_baseline': [cpu_case(64, 1), cpu_case(4096, cores), cpu_case(8192, cores, v1=True)]}
    text = json.dumps(res, indent=1)
    print(text)
    if len(sys.argv) > 1 and not sys.argv[1].startswith('-'):
        open(sys.argv[1], 'w').write(text)


if __name__ == '__main__':
    main()
