"""`suite: dflex` — factory for the DiffRL / dflex environments, same contract as mineral/envs/dflex.py:6-39.
The simulator is an optional dependency: importing this module is free, `make_envs` needs `dflex` installed."""

DEFAULT_KWARGS = {
    'ant': ('AntEnv', 1000, 16),
    'cartpoleswingup': ('CartPoleSwingUpEnv', 240, 4),
    'cheetah': ('CheetahEnv', 1000, 16),
    'hopper': ('HopperEnv', 1000, 16),
    'humanoid': ('HumanoidEnv', 1000, 48),
    'snu_humanoid': ('SNUHumanoidEnv', 1000, 8),
}


def make_envs(config):
    kwargs = dict(config.task.env.to_container())
    name, num_envs = kwargs.pop('env_name'), kwargs.pop('numEnvs')
    cls_name, episode_length, mm_caching = DEFAULT_KWARGS[name]
    kwargs = {'episode_length': episode_length, 'MM_caching_frequency': mm_caching, **kwargs}
    try:
        import dflex.envs as dflex_envs
    except ImportError as e:
        raise ImportError('task.suite=dflex needs the `dflex` package (DiffRL); it is not installed') from e
    return getattr(dflex_envs, cls_name)(num_envs=num_envs, device=config.sim_device, seed=config.seed, **kwargs)
