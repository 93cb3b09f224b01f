"""CPU restatement of the critic side of the reference's SHAC agent — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(mineral_amd/) never does.  Pinned by tests/golden/g13_shac_critic.npz, which tests/golden/make_golden_shac.py recorded
by calling the reference's own `SHAC.compute_target_values`, `CriticDataset` and `SHAC.update_critic` here.

Follows mineral/agents/diffrl/shac.py:824-889 (update_critic, compute_critic_loss, compute_target_values),
mineral/agents/diffrl/utils.py:25-69 (soft_update, grad_norm, CriticDataset) and mineral/agents/diffrl/models.py:149-256
(Critic / EnsembleCritic: MLP towers ending in a plain Linear to 1).
"""

from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ppo_oracle as P
from .sac_oracle import TowerCfg


def target_values(rew, done_mask, next_values, gamma, lam, method='td-lambda'):
    """shac.py:876-889.  All inputs (T, N) fp32; returns (T, N)."""
    if method == 'one-step':
        return rew + gamma * next_values
    T, N = rew.shape
    Ai, Bi, lm = torch.zeros(N), torch.zeros(N), torch.ones(N)
    out = torch.zeros(T, N)
    for i in reversed(range(T)):
        lm = lm * lam * (1.0 - done_mask[i]) + done_mask[i]
        adjusted = (1.0 - lm) / (1.0 - lam) * rew[i]
        Ai = (1.0 - done_mask[i]) * (lam * gamma * Ai + gamma * next_values[i] + adjusted)
        Bi = gamma * (next_values[i] * done_mask[i] + Bi * (1.0 - done_mask[i])) + rew[i]
        out[i] = (1.0 - lam) * Ai + lm * Bi
    return out


def target_values_with_entropy(rew, done_mask, next_values, entropy, alpha, gamma, lam, method='td-lambda', offset_by_target_entropy=False,
                               scale_by_target_entropy=False, target_entropy=None):
    """shac.py:893-916: the entropy bonus alpha * entropy joins the reward before the scan; `entropy` is -logprobs or the
    distribution entropy, (T, N)."""
    if offset_by_target_entropy:
        entropy = (entropy + abs(target_entropy)) * 0.5
    if scale_by_target_entropy:
        entropy = entropy * (1.0 / abs(target_entropy))
    if method == 'one-step':
        return (rew + alpha * entropy) + gamma * next_values
    T, N = rew.shape
    Ai, Bi, lm = torch.zeros(N), torch.zeros(N), torch.ones(N)
    out = torch.zeros(T, N)
    for i in reversed(range(T)):
        lm = lm * lam * (1.0 - done_mask[i]) + done_mask[i]
        r = rew[i] + alpha * entropy[i]
        adjusted = (1.0 - lm) / (1.0 - lam) * r
        Ai = (1.0 - done_mask[i]) * (lam * gamma * Ai + gamma * next_values[i] + adjusted)
        Bi = gamma * (next_values[i] * done_mask[i] + Bi * (1.0 - done_mask[i])) + r
        out[i] = (1.0 - lam) * Ai + lm * Bi
    return out


def dataset_batches(obs, targets, batch_size):
    """CriticDataset without shuffle (utils.py:39-69): (T, N, ...) flattened time-major, contiguous slices, last one short."""
    o, t = obs.reshape(-1, obs.shape[-1]), targets.reshape(-1)
    n = t.shape[0]
    return [(o[s:min(s + batch_size, n)], t[s:min(s + batch_size, n)]) for s in range(0, n, batch_size)]


def critic_prefixes(n_critics, single=False):
    """Parameter-name prefixes: `critics.<c>` for EnsembleCritic (models.py:197-214), `critic_mlp` for Critic (models.py:158)."""
    return ['critic_mlp'] if single else [f'critics.{c}' for c in range(n_critics)]


def critic_shapes(obs_dim, units, n_critics, single=False):
    res = []
    for pre in critic_prefixes(n_critics, single):
        in_f = obs_dim
        for i, u in enumerate(units):
            b = 3 * i
            res += [(f'{pre}.mlp.{b}.weight', (u, in_f)), (f'{pre}.mlp.{b}.bias', (u,)), (f'{pre}.mlp.{b + 1}.weight', (u,)),
                    (f'{pre}.mlp.{b + 1}.bias', (u,))]
            in_f = u
        b = 3 * len(units)
        res += [(f'{pre}.mlp.{b}.weight', (1, in_f)), (f'{pre}.mlp.{b}.bias', (1,))]
    return res


def critic_values(pc, obs, units, n_critics, act_type='SiLU', single=False):
    """EnsembleCritic.forward(return_type='all') (models.py:236-243): (n_critics, rows, 1); with single=True the one tower of
    Critic.forward (models.py:169-178), stacked the same way."""
    cfg = TowerCfg(units, act_type=act_type)
    out = []
    for pre in critic_prefixes(n_critics, single):
        h = P.mlp_forward(pc, pre, obs, cfg)
        b = 3 * len(units)
        out.append(F.linear(h, pc[f'{pre}.mlp.{b}.weight'], pc[f'{pre}.mlp.{b}.bias']))
    return torch.stack(out)


def critic_loss(pc, obs, target, units, n_critics, act_type='SiLU', single=False):
    """compute_critic_loss (shac.py:868-874): mean over critics of the per-critic MSE."""
    vs = critic_values(pc, obs, units, n_critics, act_type, single)
    return torch.stack([F.mse_loss(v.squeeze(-1), target) for v in vs]).mean()


def update_critic(pc, opt_state, batches, units, n_critics, iterations, lr, optim_type='AdamW', betas=(0.7, 0.95), max_grad_norm=0.5,
                  truncate_grads=True, act_type='SiLU', single=False):
    """shac.py:827-866.  Returns per-iteration value_loss and mean gradient norms before / after the clip."""
    res = {'value_loss': [], 'grad_norm_before_clip/critic': [], 'grad_norm_after_clip/critic': []}
    for _ in range(iterations):
        total, before, after = 0.0, [], []
        for obs, tgt in batches:
            leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in pc.items())
            loss = critic_loss(leaves, obs, tgt, units, n_critics, act_type, single)
            grads = [g.nan_to_num(0.0, 0.0, 0.0) for g in torch.autograd.grad(loss, list(leaves.values()))]
            if truncate_grads:
                before.append(torch.sqrt(sum(torch.sum(g ** 2) for g in grads)))
                if max_grad_norm is not None:
                    grads, _ = P.clip_grad_norm(grads, max_grad_norm)
                after.append(torch.sqrt(sum(torch.sum(g ** 2) for g in grads)))
            P.adam_step(pc, OrderedDict(zip(leaves.keys(), grads)), opt_state, lr, optim_type, betas=betas)
            total = total + loss.detach()
        res['value_loss'].append(total / len(batches))
        res['grad_norm_before_clip/critic'].append(torch.mean(torch.stack(before)))
        res['grad_norm_after_clip/critic'].append(torch.mean(torch.stack(after)))
    return res


def soft_update(p, p_target, alpha):
    """utils.py:25-28: target = alpha * target + (1 - alpha) * current (note the convention)."""
    with torch.no_grad():
        for k in p_target:
            p_target[k].mul_(alpha)
            p_target[k].add_((1.0 - alpha) * p[k])
