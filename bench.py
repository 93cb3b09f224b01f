"""PPO env-steps/sec on the synthetic (obs=64, act=12) vectorised env — the metric of BASELINE.json.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" is one full PPO iteration of the hot path: `play_steps` (T=32 policy/env steps into the device rollout
buffer, bootstrap forward, GAE, advantage / value normalisation) plus `train_epoch` (5 mini-epochs x B/mb optimizer
steps), hyper-parameters of cfgs/agent/DFlexAntPPO.yaml with minibatch_size 32768, 65536 envs PER GPU (weak
scaling: 8 GPUs = the 524288-env configuration).  Rank 0 prints one JSON line.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_HORIZON = 32
UNITS = [512, 256, 128]
OBS, ACT = 64, 12
# algorithmic work of the MLP towers per env-step: forward MACs/sample 396,416; backward 727,296; per env-step
# 1 rollout forward + 5 mini-epochs x (forward + backward) = 6,014,976 MAC = 12.03 MFLOP
FWD_MAC = 2 * (OBS * 512 + 512 * 256 + 256 * 128) + 128 * 2 * ACT + 128
BWD_MAC = 2 * FWD_MAC - 2 * OBS * 512


def make_cfg(num_envs, minibatch, device, multi_gpu, max_steps=10**12):
    from mineral_amd.config import Config
    ppo = dict(multi_gpu=multi_gpu, num_actors=num_envs, reward_shaper=dict(fn='scale', scale=1.0), max_agent_steps=max_steps,
               horizon_len=T_HORIZON, minibatch_size=minibatch, normalize_value=True, value_bootstrap=True, clip_value_loss=False,
               normalize_advantage=True, gamma=0.99, tau=0.95, entropy_coef=0.0, e_clip=0.2, use_smooth_clamp=False,
               critic_coef=4, bounds_loss_coef=0.0001, bounds_type='bound', mini_epochs=5, optim_type='AdamW',
               optim_kwargs=dict(lr=5e-4, eps=1e-5), lr_schedule='kl', kl_threshold=0.008, max_grad_norm=0.5, truncate_grads=True)
    return Config.wrap(dict(
        rl_device=device, sim_device=device, multi_gpu=multi_gpu, env_render=False, run='train', seed=42,
        task=dict(name='Synthetic', suite='synthetic', env=dict(numEnvs=num_envs, obs_dim=OBS, act_dim=ACT, episode_length=1000),
                  env_autoresets=True),
        agent=dict(algo='PPO', print_every=-1, ckpt_every=-1, tracker_len=100, metrics_kwargs={},
                   network=dict(normalize_input=True, actor_critic='ActorCritic',
                                actor_critic_kwargs=dict(separate_value_mlp=True,
                                                         mlp_kwargs=dict(units=UNITS, norm_type='LayerNorm', act_type='SiLU'),
                                                         fixed_sigma=False,
                                                         actor_dist_kwargs=dict(dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0))),
                   ppo=ppo)))


def _cpu_leg(num_envs, minibatch, threads):
    import torch
    from mineral_amd.envs.synthetic import SyntheticEnv
    from oracle import ppo_oracle as O
    torch.set_num_threads(threads)
    cfg = O.NetConfig(OBS, ACT, UNITS, 'LayerNorm', 'SiLU', False, 'dreamerv3_normal', 0.1, 1.0)
    params = O.init_params(cfg, torch.Generator().manual_seed(0))
    hp = O.PPOHParams(minibatch_size=min(minibatch, num_envs * T_HORIZON))
    env = SyntheticEnv(num_envs, OBS, ACT, 1000, device='cpu', seed=1)
    obs = env.reset()
    obs_rms, value_rms = O.rms_init((OBS,)), O.rms_init((1,))
    st = O.adam_state_init(params)
    sched = O.AdaptiveScheduler(hp.kl_threshold, hp.min_lr, hp.max_lr)
    t0 = time.perf_counter()
    noise = torch.randn(T_HORIZON, num_envs, ACT)
    data, value_rms, obs = O.play_steps(params, obs_rms, value_rms, env, obs, noise, cfg, hp)
    O.train_epoch(params, st, obs_rms, data, cfg, hp, hp.lr, sched)
    dt = time.perf_counter() - t0
    return num_envs * T_HORIZON / dt, dt, hp.minibatch_size


def cpu_baseline():
    """The CPU oracle (a torch-CPU restatement of the reference, pinned to its golden vectors) on this host's cores: one full
    PPO iteration on a bounded sample of the same workload, twice -- on every core this process may use, and on one thread,
    which is how the reference's launcher runs torch on the CPU (OMP / MKL threads pinned to 1, mineral/scripts/run.py:53-57)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    threads = max(1, min(usable, int(os.environ.get('OMP_NUM_THREADS', usable))))
    v_all, dt_all, mbs = _cpu_leg(16384, 32768, threads)
    v_one, dt_one, _ = _cpu_leg(2048, 32768, 1)
    return {'value': v_all, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'cores_machine': os.cpu_count(), 'cores_usable': usable,
            'sample': f'1 PPO iteration, 16384 envs x T={T_HORIZON}, minibatch {mbs}, 5 mini-epochs, torch-CPU oracle on {threads} threads, {dt_all:.1f} s',
            'single_thread': {'value': v_one, 'unit': 'env-steps/s', 'cores': 1,
                              'sample': f'1 PPO iteration, 2048 envs x T={T_HORIZON}, one minibatch of 65536 rows, 5 mini-epochs, 1 thread '
                                        f'(the reference launcher pins torch to one CPU thread), {dt_one:.1f} s'}}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` at this workload, from the committed rocprofv3 counter passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs by tools/pmc_traffic.sh and cannot be read from inside this process);
    None when no measurement of that kernel is on file."""
    path = os.path.join(ROOT, 'profiles', 'r03_traffic.json')
    try:
        with open(path) as f:
            kernels = json.load(f)['kernels']
    except (OSError, KeyError, ValueError):
        return None
    if kernel in kernels:
        return kernels[kernel]['hbm_bytes']
    # a timing class may cover several instantiations (dgrad_split_kernel<true>: the launches with ordinary and with
    # non-temporal output stores, <true, false> and <true, true>): the launch-weighted mean of those on file
    if kernel.endswith('>'):
        fam = [v for k, v in kernels.items() if k.startswith(kernel[:-1] + ',')]
        n = sum(v['launches'] for v in fam)
        if n:
            return sum(v['hbm_bytes'] * v['launches'] for v in fam) / n
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--num-envs', type=int, default=65536, help='envs per GPU')
    ap.add_argument('--minibatch', type=int, default=32768, help='minibatch rows per GPU')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo = rehearsal of the multi-rank path through host memory (ranks may share a GPU)')
    ap.add_argument('--force-dist', action='store_true',
                    help='run the multi-rank code path (process group, collectives) even with one rank: an RCCL rehearsal on a 1-GPU box')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the run at the shipped minibatch_size (2048)')
    args = ap.parse_args()

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    multi = world > 1 or args.force_dist
    if args.backend == 'gloo':
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = f'cuda:{local_rank}'
    if multi:
        import torch.distributed as dist
        from mineral_amd import parallel
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(dev))
        else:
            dist.init_process_group('gloo')
        coll = parallel.collectives(dist)

    from mineral_amd import _hip
    from mineral_amd.agents import PPO
    from mineral_amd.envs import synthetic

    cfg = make_cfg(args.num_envs, args.minibatch, dev, multi)
    cfg.env_seed = 365838 + rank          # launcher sub-seed 'env_train' of seed 42, offset by rank
    torch.manual_seed(671155 + rank)      # 'train' sub-seed (action noise)
    env = synthetic.make_envs(cfg)
    logdir = os.path.join('/tmp', f'mrl_bench_{os.getpid()}')
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the agent prints its model like the reference does; stdout carries the JSON line only
        agent = PPO(cfg, logdir=logdir, accelerator=None, datasets=None, env=env)
    agent.obs = agent._convert_obs(env.reset())

    def one_step():
        agent.epoch += 1
        agent.play_steps()
        agent.train_epoch()

    def fence():
        torch.cuda.synchronize()
        if multi:
            coll.barrier()
        torch.cuda.synchronize()

    L = _hip.lib()
    timing = not args.no_kernel_timing
    if timing:
        L.mrl_profile_enable(1)           # warm-up: time every GEMM instantiation to find the dominant one
    for _ in range(args.warmup):
        one_step()
    if timing:
        torch.cuda.synchronize()
        L.mrl_profile_enable(2)           # fixes the dominant instantiation from the warm-up's timings ...
        L.mrl_profile_enable(0)           # ... timed region: no events, the product's own launch order
    calls0 = dict(agent._dist.calls) if multi and hasattr(agent._dist, 'calls') else None
    fence()
    t0 = time.perf_counter()
    step_times = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        one_step()
        torch.cuda.synchronize()     # per-update times for the median; the update itself already syncs once per mini-epoch (KL-adaptive lr)
        step_times.append(time.perf_counter() - ts)
    fence()
    dt = time.perf_counter() - t0
    collectives_per_update = None
    if calls0 is not None:
        collectives_per_update = {k: (v - calls0.get(k, 0)) / args.steps for k, v in agent._dist.calls.items()}
    flat64 = agent.model.flat.double()
    param_checksum = [float(flat64.sum().item()), float((flat64 * flat64).sum().item())]   # run-to-run / rank-count comparisons
    import hashlib
    param_sha256 = hashlib.sha256(agent.model.flat.detach().cpu().numpy().tobytes()).hexdigest()   # every bit of every parameter
    dominant, dt_events = None, None
    if timing:
        # the same K steps once more with HIP events around every launch of the dominant kernel, on the stream it is launched
        # on.  With events on, the library issues every launch in order, so a launch's duration is its own; in the timed region
        # above the weight-gradient launches run beside the dgrad launches (any-order launches).
        L.mrl_profile_enable(2 if args.warmup > 0 else 1)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        fence()
        dt_events = time.perf_counter() - t1
        if rank == 0:
            import ctypes as C
            ms_, n_, flops_ = C.c_double(0), C.c_int64(0), C.c_double(0)
            name_ = C.create_string_buffer(128)
            L.mrl_profile_read(C.byref(ms_), C.byref(n_), C.byref(flops_), name_, 128)
            dominant = (ms_.value, n_.value, flops_.value, name_.value.decode())
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        coll.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # every rank must hold the same parameters after the same number of optimizer steps
        sums = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
        coll.all_gather(sums, torch.tensor(param_checksum, dtype=torch.float64, device=dev))
        in_sync = all(bool(torch.equal(sums[0], x)) for x in sums[1:])
        # what the process group itself reports: every rank contributes its rank and its device index
        ids = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
        coll.all_gather(ids, torch.tensor([float(dist.get_rank()), float(local_rank)], dtype=torch.float64, device=dev))
        ranks_seen = sorted(int(x[0].item()) for x in ids)
        devices_seen = [int(x[1].item()) for x in ids]

    kernel_table = None
    if timing and rank == 0:
        # one more update outside the timed region with events around every instantiation: the per-kernel table
        import ctypes as C
        L.mrl_profile_enable(1)
        one_step()
        torch.cuda.synchronize()
        text = C.create_string_buffer(16384)
        n = L.mrl_profile_read_all(text, 16384)
        kernel_table = []
        for line in text.value.decode().splitlines() if n > 0 else []:
            ms, launches, flops, kname = line.split('\t')
            ms, launches, flops = float(ms), int(launches), float(flops)
            kernel_table.append({'name': kname, 'launches': launches, 'avg_us': 1e3 * ms / launches, 'total_ms': ms,
                                 'tflops': (flops / (ms * 1e-3) / 1e12) if flops > 0 else None})
        L.mrl_profile_enable(0)
    elif timing:
        one_step()   # the other ranks take part in the extra update's collectives

    env_steps = args.num_envs * T_HORIZON * world * args.steps
    out = {
        'metric': f'PPO env-steps/sec, synthetic env (obs=64, act=12), {args.num_envs} envs per MI355X',
        'value': env_steps / dt, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps, 'ms_per_step_median': 1e3 * sorted(step_times)[len(step_times) // 2], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'PPO DFlexAntPPO hparams, {args.num_envs} envs/GPU x T=32, minibatch {args.minibatch}, '
                               f'5 mini-epochs, MLP [512,256,128] LayerNorm+SiLU, AdamW, KL-adaptive lr',
                   'arithmetic': 'fp32 storage and accumulation; the GEMMs of the hidden layers multiply exact three-term bf16 splits of '
                                 'their fp32 operands on the bf16 matrix cores (error below that of an fp32 fma chain)',
                   'num_envs_total': args.num_envs * world, 'ms_per_update': 1e3 * dt / args.steps,
                   'parallelism': f'dp{world} (envs sharded, 1 RCCL all-reduce of the flat gradient per optimizer step)'},
    }
    if collectives_per_update is not None:
        out['collectives_per_update'] = collectives_per_update
    out['param_checksum'] = param_checksum
    out['param_sha256'] = param_sha256
    if multi:
        out['ranks_in_sync'] = in_sync
        out['rccl'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'ranks_seen': ranks_seen,
                       'devices_seen': devices_seen}
    if rank == 0:
        if timing and dominant is not None:
            class _V:   # the dominant kernel of the timed region, read before the extra table-gathering update
                def __init__(self, v):
                    self.value = v
            ms, n, flops, kname = _V(dominant[0]), _V(dominant[1]), _V(dominant[2]), dominant[3]
            if n.value > 0:
                achieved = flops.value / (ms.value * 1e-3) / 1e12
                # kernels that multiply exact three-term bf16 splits of the fp32 operands issue six bf16 MFMA terms per
                # fp32 product: their ceiling for algorithmic (fp32) FLOPs is the dense bf16 peak / 6; the others run the
                # fp32 MFMA (157.3 TFLOP/s)
                split = 'split_kernel' in kname or (kname.startswith('mrl::gemm_kernel<') and kname.count(',') == 10 and (kname.endswith(', 1>') or kname.endswith(', 2>')))
                peak = 2500.0 / 6.0 if split else 157.3
                out['roofline'] = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                                   'traffic': measured_traffic(kname), 'kernel': kname, 'launches': int(n.value),
                                   'avg_launch_us': 1e3 * ms.value / n.value,
                                   'peak_basis': ('dense bf16 MFMA peak 2500 / 6 split terms per fp32 product' if split else 'dense fp32 MFMA peak'),
                                   'frac_of_fp32_mfma_peak': achieved / 157.3,
                                   'kernel_instances': (['mrl::dgrad_split_kernel<true, false>', 'mrl::dgrad_split_kernel<true, true>'] if kname == 'mrl::dgrad_split_kernel<true>' else [kname]),
                                   'measured_over': f'a second pass of the same {args.steps} steps with HIP events around every launch of this kernel '
                                                    f'(every launch in order), {1e3 * dt_events / args.steps:.1f} ms per step'}
                # the whole update against the same ceiling: algorithmic MACs per env step = the rollout's forward + 5 mini-epochs x
                # (forward + backward) of both towers and heads, less the first layers' input gradients, which nothing needs
                w_all = 2 * (64 * 512 + 512 * 256 + 256 * 128) + 128 * 24 + 128 * 1
                flop_per_env_step = 2.0 * (16 * w_all - 5 * 2 * 64 * 512)
                upd_tflops = flop_per_env_step * args.num_envs * T_HORIZON / (dt / args.steps) / 1e12
                out['roofline']['update_tflops'] = upd_tflops
                out['roofline']['update_frac'] = upd_tflops / peak
                if kernel_table:
                    out['roofline']['kernels'] = kernel_table[:12]
                    # the streaming kernels of the update: HBM bytes per launch from the committed counter passes over this same
                    # workload (profiles/r03_traffic.json) over the event time measured here
                    hbm = []
                    for row in kernel_table:
                        nm = row['name'].split(' + ')[0]
                        if not any(k in nm for k in ('gae_kernel', 'gae_normalize', 'moments_partial', 'rollout_write', 'ln_bwd', 'lnbwd_wgrad0', 'lnfwd0', 'ppo_tail', 'reduce_slabs',
                                                     'reduce_flat', 'gradnorm', 'adam')):
                            continue
                        by = measured_traffic(nm if 'ln_bwd' in nm else nm.split('<')[0])
                        if by is None and 'lnbwd_wgrad0' in nm:
                            by = measured_traffic('mrl::lnbwd_wgrad0_kernel<64, false>')
                        if by is None and 'lnfwd0' in nm:
                            by = measured_traffic('mrl::lnfwd0_kernel')
                        if by is None and 'ppo_tail' in nm:
                            by = measured_traffic('mrl::ppo_tail_kernel<false>')
                        if by is None and nm == 'moments_partial_kernel':
                            by = measured_traffic('moments_partial4_kernel')   # the 16-byte-load variant taken for column counts that are multiples of 4
                        if 'adam' in row['name'] and by is not None:
                            by2 = measured_traffic('adam_kernel')
                            by = by + by2 if by2 is not None else by
                        hbm.append({'name': row['name'], 'launches': row['launches'], 'avg_us': row['avg_us'], 'hbm_bytes': by,
                                    'achieved_gbs': (by / (row['avg_us'] * 1e-6) / 1e9) if by else None,
                                    'frac_of_8tbs': (by / (row['avg_us'] * 1e-6) / 8e12) if by else None})
                    out['roofline']['hbm_kernels'] = hbm
        if not args.no_secondary and not multi and args.minibatch != 2048:
            # the minibatch size the reference ships for this network (cfgs/agent/DFlexAntPPO.yaml: minibatch_size 2048):
            # 5,120 optimizer steps per update instead of 320, a launch- and latency-bound regime
            del agent
            torch.cuda.empty_cache()
            cfg2 = make_cfg(args.num_envs, 2048, dev, False)
            cfg2.env_seed = 365838
            env2 = synthetic.make_envs(cfg2)
            with contextlib.redirect_stdout(sys.stderr):
                agent2 = PPO(cfg2, logdir=logdir, accelerator=None, datasets=None, env=env2)
            agent2.obs = agent2._convert_obs(env2.reset())

            def step2():
                agent2.epoch += 1
                agent2.play_steps()
                agent2.train_epoch()
            step2()
            torch.cuda.synchronize()
            n2 = 5
            times2 = []
            for _ in range(n2):
                t2 = time.perf_counter()
                step2()
                torch.cuda.synchronize()
                times2.append(time.perf_counter() - t2)
            dt2 = sum(times2) / n2
            out['secondary'] = {'minibatch': 2048, 'optimizer_steps_per_update': 5 * args.num_envs * T_HORIZON // 2048, 'updates_timed': n2,
                                'ms_per_update': 1e3 * dt2, 'ms_per_update_median': 1e3 * sorted(times2)[n2 // 2],
                                'ms_per_update_all': [round(1e3 * t, 2) for t in times2],
                                'value': args.num_envs * T_HORIZON / dt2, 'unit': 'env-steps/s'}
        if not args.no_cpu_baseline and not multi:   # the CPU leg is timed at N=1 only
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if multi:
        coll.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
