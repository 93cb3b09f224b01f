"""cProfile of the host side of ten SAC epochs at the benchmark configuration of tools/sac_bench.py.
Run on the GPU box: python tools/host_profile_sac.py"""
import cProfile, pstats, sys, os, io
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import sac_bench as sb
from mineral_amd.agents import SAC
from mineral_amd.envs import synthetic
cfg = sb.make_cfg('cuda:0'); cfg.env_seed = 1
env = synthetic.make_envs(cfg)
agent = SAC(cfg, logdir='/tmp/mrl_sac_bench', accelerator=None, datasets=None, env=env)
agent._checkpoint_save = lambda *a, **k: None
agent.obs = agent._convert_obs(env.reset()); agent.set_eval()
traj, steps = agent.explore_env(env, cfg.agent.sac.warm_up, random=True)
agent.memory.add_to_buffer(traj)
def one_epoch():
    agent.epoch += 1
    traj, _ = agent.explore_env(env, sb.T, sample=True)
    agent.memory.add_to_buffer(traj)
    return agent.update_net(agent.memory)
for _ in range(3): one_epoch()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): one_epoch()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40); print(s.getvalue()[:8000])
