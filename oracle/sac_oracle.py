"""CPU restatement of the reference's SAC update loop — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(mineral_amd/) never does.  Pinned by tests/golden/g12_sac_*.npz, which tests/golden/make_golden_sac.py recorded from
the reference's own `SAC.train()` run here with injected noise / random actions / replay indices.

Functional torch-CPU code (explicit parameter dicts, explicit optimizer state, explicit noise), following
mineral/agents/sac/sac.py, mineral/agents/ddpg/models.py, mineral/nets/dist.py:39-43 (squashed_normal) and
mineral/nets/distributions.py:10-52 (tanh transform and its log-det-Jacobian).  Network: Actor = MLP tower + `mu` and
`sigma` heads; EnsembleQ = n_critics towers on cat(obs, action) ending in a plain Linear to 1.
"""

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ppo_oracle as P


class TowerCfg:
    """What ppo_oracle.mlp_forward needs to run one [Linear -> LayerNorm -> SiLU] tower."""

    def __init__(self, units, norm_type='LayerNorm', act_type='SiLU', ln_eps=1e-5):
        self.units, self.norm_type, self.act_type, self.ln_eps = list(units), norm_type, act_type, ln_eps
        self.layer_stride = 1 + int(norm_type is not None) + 1


class SACConfig:
    def __init__(self, obs_dim, act_dim, actor_units, critic_units, n_critics=2, minlogstd=-5.0, maxlogstd=2.0, actor_kind=('LayerNorm', 'SiLU'),
                 critic_kind=('LayerNorm', 'SiLU')):
        self.obs_dim, self.act_dim, self.n_critics = obs_dim, act_dim, n_critics
        self.actor, self.critic = TowerCfg(actor_units, *actor_kind), TowerCfg(critic_units, *critic_kind)   # (norm_type, act_type) of the mlp_kwargs
        self.minlogstd, self.maxlogstd = minlogstd, maxlogstd


class SACHParams:
    """cfgs/agent/DFlexAntSAC.yaml:40-66 defaults."""

    def __init__(self, **kw):
        self.batch_size, self.tau, self.nstep, self.gamma = 2048, 0.01, 3, 0.99
        self.mini_epochs, self.update_actor_interval, self.update_targets_interval = 8, 1, 1
        self.init_alpha, self.alpha_lr, self.actor_lr, self.critic_lr = 1.0, 5e-3, 5e-4, 5e-4
        self.backup_entropy, self.optim_type, self.max_grad_norm = True, 'AdamW', 0.5
        self.target_entropy_scalar = 1.0
        self.rms_eps = 1e-4
        self.__dict__.update(kw)


def actor_shapes(cfg):
    res, in_f = [], cfg.obs_dim
    for i, u in enumerate(cfg.actor.units):
        b = i * cfg.actor.layer_stride
        res += [(f'actor_mlp.mlp.{b}.weight', (u, in_f)), (f'actor_mlp.mlp.{b}.bias', (u,))]
        if cfg.actor.norm_type is not None:
            res += [(f'actor_mlp.mlp.{b + 1}.weight', (u,)), (f'actor_mlp.mlp.{b + 1}.bias', (u,))]
        in_f = u
    return res + [('mu.weight', (cfg.act_dim, in_f)), ('mu.bias', (cfg.act_dim,)), ('sigma.weight', (cfg.act_dim, in_f)), ('sigma.bias', (cfg.act_dim,))]


def critic_shapes(cfg):
    """state_dict order of EnsembleQ (ddpg/models.py:72-110): critics.{i}.mlp.{...}, the last module a plain Linear."""
    res = []
    for c in range(cfg.n_critics):
        in_f = cfg.obs_dim + cfg.act_dim
        for i, u in enumerate(cfg.critic.units):
            b = i * cfg.critic.layer_stride
            res += [(f'critics.{c}.mlp.{b}.weight', (u, in_f)), (f'critics.{c}.mlp.{b}.bias', (u,))]
            if cfg.critic.norm_type is not None:
                res += [(f'critics.{c}.mlp.{b + 1}.weight', (u,)), (f'critics.{c}.mlp.{b + 1}.bias', (u,))]
            in_f = u
        b = cfg.critic.layer_stride * len(cfg.critic.units)
        res += [(f'critics.{c}.mlp.{b}.weight', (1, in_f)), (f'critics.{c}.mlp.{b}.bias', (1,))]
    return res


# ---- normaliser variant of SAC (sac.py:34: fp32 state, eps 1e-4, initial count = eps, clamp +-5) ----------------------
def rms_init(shape, eps=1e-4):
    return P.rms_init(shape, initial_count=eps, dtype=torch.float32)


def rms_normalize(state, x, eps=1e-4):
    return P.rms_normalize(state, x, eps=eps, clamp=(-5.0, 5.0))


# ---- networks ----------------------------------------------------------------------------------------------------------
def actor_raw(pa, x, cfg):
    h = P.mlp_forward(pa, 'actor_mlp', x, cfg.actor)
    return F.linear(h, pa['mu.weight'], pa['mu.bias']), F.linear(h, pa['sigma.weight'], pa['sigma.bias'])


def squashed_sample(mu, logstd_raw, eps, cfg):
    """dist.py:39-43 + distributions.py:38-52: u = mu + sigma*eps, a = tanh(u); log-prob of a from the cached u:
    sum_a [ log N(u; mu, sigma) - 2*(log 2 - u - softplus(-2u)) ]  (keepdim), as `get_actions(logprob=True)` at sac.py:132-134."""
    logstd = torch.clamp(logstd_raw, cfg.minlogstd, cfg.maxlogstd)
    sigma = logstd.exp()
    u = mu + sigma * eps
    a = u.tanh()
    base = -((u - mu) ** 2) / (2 * sigma ** 2) - sigma.log() - math.log(math.sqrt(2 * math.pi))
    logdet = 2.0 * (math.log(2.0) - u - F.softplus(-2.0 * u))
    return a, (base - logdet).sum(-1, keepdim=True), sigma


def q_values(pc, x, a, cfg):
    """ddpg/models.py:102-110: every critic on cat(state, action)."""
    inp = torch.cat((x, a), dim=1)
    out = []
    for c in range(cfg.n_critics):
        h = P.mlp_forward(pc, f'critics.{c}', inp, cfg.critic)
        b = cfg.critic.layer_stride * len(cfg.critic.units)
        out.append(F.linear(h, pc[f'critics.{c}.mlp.{b}.weight'], pc[f'critics.{c}.mlp.{b}.bias']))
    return out


def q_min(pc, x, a, cfg):
    return torch.min(torch.stack(q_values(pc, x, a, cfg)), dim=0).values


def _step(params, loss, leaves, opt_state, lr, hp):
    """optimizer_update (sac.py:336-347): backward, clip_grad_norm_ over the optimizer's parameters, AdamW step."""
    grads = torch.autograd.grad(loss, list(leaves.values()))
    gnorm = None
    if hp.max_grad_norm is not None:
        grads, gnorm = P.clip_grad_norm(list(grads), hp.max_grad_norm)
    P.adam_step(params, OrderedDict(zip(leaves.keys(), grads)), opt_state, lr, hp.optim_type)
    return gnorm


def _leaves(params):
    return OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())


def update_critic(pa, pc, pt, log_alpha, opt_c, obs_rms, batch, eps_next, cfg, hp, normalize_input=True):
    """sac.py:287-313."""
    obs, action, reward, next_obs, done = batch
    with torch.no_grad():
        nx = rms_normalize(obs_rms, next_obs, hp.rms_eps) if normalize_input else next_obs
        mu, ls = actor_raw(pa, nx, cfg)
        a2, logp2, _ = squashed_sample(mu, ls, eps_next, cfg)
        tq = q_min(pt, nx, a2, cfg)
        if hp.backup_entropy:
            tq = tq - log_alpha.exp() * logp2
        target = reward + (1 - done) * (hp.gamma ** hp.nstep) * tq
    x = rms_normalize(obs_rms, obs, hp.rms_eps) if normalize_input else obs
    leaves = _leaves(pc)
    qs = q_values(leaves, x, action, cfg)
    loss = torch.sum(torch.stack([F.mse_loss(q, target) for q in qs]))
    gnorm = _step(pc, loss, leaves, opt_c, hp.critic_lr, hp)
    return loss.detach(), gnorm, target


def update_actor(pa, pc, log_alpha, opt_a, opt_alpha, obs_rms, obs, eps, cfg, hp, normalize_input=True):
    """sac.py:315-334; `log_alpha` is a 0-dim tensor updated in place."""
    x = rms_normalize(obs_rms, obs, hp.rms_eps) if normalize_input else obs
    leaves = _leaves(pa)
    mu, ls = actor_raw(leaves, x, cfg)
    a, logp, _ = squashed_sample(mu, ls, eps, cfg)
    q = q_min(pc, x, a, cfg)
    alpha = log_alpha.detach().exp()
    loss = (alpha * logp - q).mean()
    gnorm = _step(pa, loss, leaves, opt_a, hp.actor_lr, hp)
    entropy = -logp.detach()
    target_entropy = -cfg.act_dim * hp.target_entropy_scalar
    la = {'log_alpha': log_alpha}
    leaf = _leaves(la)
    alpha_loss = (leaf['log_alpha'].exp() * (entropy - target_entropy)).mean()
    _step(la, alpha_loss, leaf, opt_alpha, hp.alpha_lr, hp)
    return loss.detach(), alpha_loss.detach(), entropy.mean(), gnorm


def soft_update(pt, pc, tau):
    """ddpg/utils.py:7-11."""
    with torch.no_grad():
        for k in pt:
            pt[k].mul_(1.0 - tau).add_(pc[k] * tau)


def update_net(pa, pc, pt, log_alpha, opts, obs_rms, sample_fn, noise_fn, cfg, hp, mini_epoch0=-1, record=None):
    """sac.py:251-275.  `sample_fn(i)` returns the batch of mini-epoch i, `noise_fn(i, which)` the standard-normal draws
    ('next' for the target action, 'actor' for the policy update).  Returns the advanced mini_epoch counter."""
    mini_epoch = mini_epoch0
    for i in range(hp.mini_epochs):
        mini_epoch += 1
        batch = sample_fn(i)
        c_loss, c_norm, target = update_critic(pa, pc, pt, log_alpha, opts['critic'], obs_rms, batch, noise_fn(i, 'next'), cfg, hp)
        rec = {'loss/critic': c_loss, 'grad_norm/critic': c_norm, 'target_mean': target.mean()}
        if mini_epoch % hp.update_actor_interval == 0:
            a_loss, al_loss, ent, a_norm = update_actor(pa, pc, log_alpha, opts['actor'], opts['alpha'], obs_rms, batch[0], noise_fn(i, 'actor'), cfg, hp)
            rec.update({'loss/actor': a_loss, 'loss/alpha': al_loss, 'entropy': ent, 'grad_norm/actor': a_norm})
        if mini_epoch % hp.update_targets_interval == 0:
            soft_update(pt, pc, hp.tau)
        if record is not None:
            record.append(rec)
    return mini_epoch


def get_actions(pa, obs_rms, obs, eps, cfg, hp, normalize_input=True):
    """sac.py:118-136 with sample=True."""
    x = rms_normalize(obs_rms, obs, hp.rms_eps) if normalize_input else obs
    mu, ls = actor_raw(pa, x, cfg)
    return squashed_sample(mu, ls, eps, cfg)[0]
