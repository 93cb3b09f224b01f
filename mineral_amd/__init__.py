"""mineral_amd — MI355X-native PPO actor-learner hot path behind mineral's agent / env plugin surface.

The compute lives in `libmineral_ppo.so` (hand-written HIP for gfx950, C ABI in include/mineral_ppo.h);
this package is the host-side mirror of the reference's Python interfaces for that path.
"""

__version__ = '0.1.0'
