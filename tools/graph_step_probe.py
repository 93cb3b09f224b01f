"""Measurement: one optimizer step of the shipped minibatch (2,048 rows) as eager launches and as a replayed HIP graph.

The step (mrl_rms_merge -> mrl_ppo_grad -> mrl_adam_step on one fixed row range) is captured once with torch.cuda.CUDAGraph
and replayed; both variants run the same kernels on the same data, so the difference is what launching costs.  Timing only:
a real update walks over the row ranges and changes the step count and the learning rate between steps.
Run on the GPU box: MRL_ANY_ORDER=0 python tools/graph_step_probe.py [minibatch]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mineral_amd import _hip  # noqa: E402
from mineral_amd.agents import PPO  # noqa: E402
from mineral_amd.envs import synthetic  # noqa: E402


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = 'cuda:0'
    torch.cuda.set_device(0)
    cfg = bench.make_cfg(4096, mb, dev, False)
    env = synthetic.make_envs(cfg)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        agent = PPO(cfg, logdir='/tmp/mrl_graph_probe', accelerator=None, datasets=None, env=env)
    agent.obs = agent._convert_obs(env.reset())
    agent.epoch += 1
    agent.play_steps()
    agent.train_epoch()           # a real update first: statistics, moments and Adam state are warm
    L = _hip.lib()
    obs_rms = agent.obs_rms['obs'].c_struct()
    net, flat, buf = agent.model.net, agent.model.flat, agent.storage

    def step(s):
        _hip.check(L.mrl_rms_merge(obs_rms, _hip.ptr(agent._mb_mean[0]), _hip.ptr(agent._mb_var[0]), float(mb), _hip.ptr(agent._obs_mean_f),
                                   _hip.ptr(agent._obs_std_f), s), 'merge')
        _hip.check(L.mrl_ppo_grad(net, _hip.ptr(flat), agent.hp, buf.c, 0, mb, _hip.ptr(agent._obs_mean_f), _hip.ptr(agent._obs_std_f),
                                  obs_rms.clamp_lo, obs_rms.clamp_hi, _hip.ptr(agent.grads), _hip.ptr(agent._metrics[0]), _hip.ptr(agent._ws),
                                  agent._ws.numel(), s), 'grad')
        _hip.check(L.mrl_adam_step(_hip.ptr(flat), _hip.ptr(agent.grads), _hip.ptr(agent.exp_avg), _hip.ptr(agent.exp_avg_sq), flat.numel(),
                                   1e-5, 0.9, 0.999, 1e-8, 0.01, 1, 1000, 0.5, 1.0, _hip.ptr(agent._gnorm), _hip.ptr(agent._adam_scratch), s), 'adam')

    n = 2000
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s = side.cuda_stream
        for _ in range(20):
            step(s)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(s)
        side.synchronize()
        eager = (time.perf_counter() - t0) / n
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step(torch.cuda.current_stream().cuda_stream)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n
    # ten steps per graph: fewer graph launches for the same kernels
    g10 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g10, stream=side):
        for _ in range(10):
            step(torch.cuda.current_stream().cuda_stream)
    for _ in range(5):
        g10.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 10):
        g10.replay()
    torch.cuda.synchronize()
    graph10 = (time.perf_counter() - t0) / n
    print(f'minibatch {mb}: eager {eager * 1e6:.1f} us per optimizer step, graph replay {graph * 1e6:.1f} us, ten steps per graph {graph10 * 1e6:.1f} us', flush=True)


if __name__ == '__main__':
    main()
