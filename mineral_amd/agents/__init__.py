"""Agent registry: `getattr(mineral_amd.agents, config.agent.algo)` as at mineral/scripts/run.py:140-144.
Third parties register with `mineral_amd.agents.MyAgent = MyAgent` (cf. examples/agents/__init__.py)."""

from .agent import Agent
from .ppo.ppo import PPO
from .sac.sac import SAC
from .template_agent import Template
