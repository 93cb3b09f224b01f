"""Env suites: `getattr(mineral_amd.envs, config.task.suite).make_envs(config)` (mineral/scripts/run.py:15-20)."""

from . import dflex, synthetic
