"""cProfile of the host side of the PPO loop at the reference's own scale (64 envs, one 2048-row minibatch): which Python /
torch calls the launch-bound regime spends its time in.  Run on the GPU box: python tools/host_profile_ppo.py"""
import cProfile, pstats, sys, os, io
sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--num-envs', '64', '--minibatch', '2048', '--no-cpu-baseline', '--no-kernel-timing']
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
