"""Learning-rate schedules of mineral/agents/ppo/utils.py (host scalars)."""

import math


class AdaptiveScheduler:
    """KL-adaptive LR: shrink by 1.5 above 2x the threshold, grow by 1.5 below half of it (utils.py:7-20)."""

    def __init__(self, kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2):
        self.kl_threshold, self.min_lr, self.max_lr = kl_threshold, min_lr, max_lr

    def update(self, current_lr, kl_dist):
        lr = current_lr
        if kl_dist > 2.0 * self.kl_threshold:
            lr = max(current_lr / 1.5, self.min_lr)
        if kl_dist < 0.5 * self.kl_threshold:
            lr = min(current_lr * 1.5, self.max_lr)
        return lr


class LinearScheduler:
    """utils.py:23-32."""

    def __init__(self, start_lr, min_lr=1e-6, max_steps=1000000):
        self.start_lr, self.min_lr, self.max_steps = start_lr, min_lr, max_steps

    def update(self, steps):
        return max(self.min_lr, self.start_lr - self.start_lr * (steps / float(self.max_steps)))


def adjust_learning_rate_cos(init_lr, epoch, mini_epochs, agent_steps, max_agent_steps):
    """utils.py:35-37."""
    return init_lr * 0.5 * (1.0 + math.cos(math.pi * (agent_steps + epoch / mini_epochs) / max_agent_steps))
