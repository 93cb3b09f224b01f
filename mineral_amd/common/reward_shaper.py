class RewardShaper:
    """Reward scaling (mineral/common/reward_shaper.py).  In the PPO rollout the multiply is fused into
    mrl_rollout_write; calling the object applies it to a tensor for other users."""

    def __init__(self, fn='scale', scale=1.0):
        if fn != 'scale':
            raise NotImplementedError(fn)
        self.fn, self.scale = fn, float(scale)

    def __call__(self, rewards):
        return rewards * self.scale
