"""Critic side of the reference's SHAC agent on the MI355X (mineral/agents/diffrl/{shac.py,models.py,utils.py}).  The actor
side back-propagates through a differentiable simulator and stays with the caller."""

from .critic import (Critic, CriticDataset, CriticLearner, EnsembleCritic, compute_target_values, compute_target_values_with_entropy, grad_norm,
                     soft_update)
