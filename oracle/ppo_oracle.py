"""CPU oracle for the PPO actor-learner hot path.

TEST INFRASTRUCTURE ONLY.  This module restates, in plain functional torch-CPU
code, the arithmetic of the reference PPO agent (jacr13/mineral) so that the
HIP kernels can be checked against it.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it; the product package
(`mineral_amd`) never does and fails loudly when its HIP library is missing.

Parity pin: every function here is checked against golden vectors that were
produced by importing the reference's own Python (see
`tests/golden/make_golden.py`, `tests/test_oracle_golden.py`).  The reference
ships no tests of its own, so those generated vectors are the only pin.

Each function cites the reference lines it follows as
`mineral/<path>:<lines>` (paths relative to the reference checkout).

Parameter naming follows the reference `state_dict`:
  actor_mlp.mlp.{0,3,6}.{weight,bias}   Linear        (with LayerNorm + act)
  actor_mlp.mlp.{1,4,7}.{weight,bias}   LayerNorm
  value_mlp.mlp.*                       same for the critic tower
  mu.{weight,bias}, sigma.{weight,bias} | sigma (fixed), value.{weight,bias}
Without a norm layer the Linear indices are {0,2,4}.
"""

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


# --------------------------------------------------------------------------
# Running mean / std  (mineral/common/normalizers.py:19-78)
# --------------------------------------------------------------------------
def rms_init(shape, initial_count=1.0, dtype=torch.float64):
    """normalizers.py:37-42 — mean 0, var 1, count `initial_count`."""
    return {
        'running_mean': torch.zeros(shape, dtype=dtype),
        'running_var': torch.ones(shape, dtype=dtype),
        'running_count': torch.tensor(initial_count, dtype=dtype),
    }


def rms_merge(state, batch_mean, batch_var, batch_count):
    """normalizers.py:44-55 — parallel-Welford merge of batch moments into the state."""
    mean, var, count = state['running_mean'], state['running_var'], state['running_count']
    delta = batch_mean - mean
    tot = count + batch_count
    new_mean = mean + delta * batch_count / tot
    m2 = var * count + batch_var * batch_count + delta**2 * count * batch_count / tot
    return {'running_mean': new_mean, 'running_var': m2 / tot, 'running_count': tot}


def rms_update(state, x, correction=1):
    """normalizers.py:57-63 — batch moments in x's dtype, merge in the state's dtype."""
    return rms_merge(state, x.mean(dim=0), x.var(dim=0, correction=correction), x.shape[0])


def _clamp(x, clamp):
    return x if clamp is None else torch.clamp(x, min=clamp[0], max=clamp[1])


def rms_normalize(state, x, eps=1e-5, clamp=(-5.0, 5.0)):
    """normalizers.py:65-68."""
    x = (x - state['running_mean'].float()) / torch.sqrt(state['running_var'].float() + eps)
    return _clamp(x, clamp)


def rms_unnormalize(state, x, eps=1e-5, clamp=(-5.0, 5.0)):
    """normalizers.py:70-73 — the clamp is applied BEFORE un-scaling."""
    x = _clamp(x, clamp)
    return torch.sqrt(state['running_var'].float() + eps) * x + state['running_mean'].float()


# --------------------------------------------------------------------------
# Network  (mineral/nets/mlp.py:25-77, mineral/nets/dist.py:31-58,
#           mineral/agents/ppo/models.py:93-151)
# --------------------------------------------------------------------------
class NetConfig:
    """Static description of the ActorCritic (models.py:30-42 defaults overridden by yaml)."""

    def __init__(self, obs_dim, act_dim, units=(512, 256, 128), norm_type='LayerNorm', act_type='SiLU',
                 fixed_sigma=False, dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0, ln_eps=1e-5):
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.units = tuple(units)
        self.norm_type, self.act_type = norm_type, act_type
        self.fixed_sigma, self.dist_type = fixed_sigma, dist_type
        self.minstd, self.maxstd, self.ln_eps = minstd, maxstd, ln_eps

    @property
    def layer_stride(self):
        # Sequential index stride between Linear layers (mlp.py:50-73): Linear, [Norm], Act
        return 1 + (self.norm_type is not None) + (self.act_type is not None)

    def param_names(self):
        names = []
        for tower in ('actor_mlp', 'value_mlp'):
            for i in range(len(self.units)):
                base = i * self.layer_stride
                names += [f'{tower}.mlp.{base}.weight', f'{tower}.mlp.{base}.bias']
                if self.norm_type is not None:
                    names += [f'{tower}.mlp.{base + 1}.weight', f'{tower}.mlp.{base + 1}.bias']
        return names


_ACTS = {
    'SiLU': F.silu,
    'ELU': F.elu,
    'ReLU': F.relu,
    'Tanh': torch.tanh,
    None: lambda x: x,
}


def init_params(cfg, generator=None):
    """Deterministic stand-in for models.py:75-91 (orthogonal init, gains sqrt2 / 1 / 0.01, zero bias).

    Parity runs inject weights, so only the shapes, names and ORDER of the
    entries (the order of `model.parameters()`, which fixes the optimizer's
    flat layout: sigma?, actor_mlp, value_mlp, value, mu, sigma) must match
    models.py:55-69.
    """
    p = OrderedDict()

    def ortho(out_f, in_f, gain):
        w = torch.empty(out_f, in_f)
        torch.nn.init.orthogonal_(w, gain=gain, generator=generator)
        return w

    if cfg.fixed_sigma:
        p['sigma'] = torch.zeros(cfg.act_dim)
    for tower in ('actor_mlp', 'value_mlp'):
        in_f = cfg.obs_dim
        for i, out_f in enumerate(cfg.units):
            base = i * cfg.layer_stride
            p[f'{tower}.mlp.{base}.weight'] = ortho(out_f, in_f, math.sqrt(2.0))
            p[f'{tower}.mlp.{base}.bias'] = torch.zeros(out_f)
            if cfg.norm_type is not None:
                p[f'{tower}.mlp.{base + 1}.weight'] = torch.ones(out_f)
                p[f'{tower}.mlp.{base + 1}.bias'] = torch.zeros(out_f)
            in_f = out_f
    p['value.weight'] = ortho(1, cfg.units[-1], 1.0)
    p['value.bias'] = torch.zeros(1)
    p['mu.weight'] = ortho(cfg.act_dim, cfg.units[-1], 0.01)
    p['mu.bias'] = torch.zeros(cfg.act_dim)
    if not cfg.fixed_sigma:
        p['sigma.weight'] = ortho(cfg.act_dim, cfg.units[-1], 0.01)
        p['sigma.bias'] = torch.zeros(cfg.act_dim)
    return p


def mlp_forward(params, tower, x, cfg):
    """mlp.py:50-77 — [Linear -> Norm -> Act] per unit."""
    act = _ACTS[cfg.act_type]
    for i, out_f in enumerate(cfg.units):
        base = i * cfg.layer_stride
        x = F.linear(x, params[f'{tower}.mlp.{base}.weight'], params[f'{tower}.mlp.{base}.bias'])
        if cfg.norm_type == 'LayerNorm':
            x = F.layer_norm(x, (out_f,), params[f'{tower}.mlp.{base + 1}.weight'],
                             params[f'{tower}.mlp.{base + 1}.bias'], cfg.ln_eps)
        elif cfg.norm_type is not None:
            raise NotImplementedError(cfg.norm_type)
        x = act(x)
    return x


def actor_critic_raw(params, obs, cfg):
    """models.py:136-151 — (mu_head, logstd, value); logstd = mu*0 + sigma."""
    xa = mlp_forward(params, 'actor_mlp', obs, cfg)
    mu = F.linear(xa, params['mu.weight'], params['mu.bias'])
    if cfg.fixed_sigma:
        sigma = params['sigma']
    else:
        sigma = F.linear(xa, params['sigma.weight'], params['sigma.bias'])
    xv = mlp_forward(params, 'value_mlp', obs, cfg)
    value = F.linear(xv, params['value.weight'], params['value.bias'])
    return mu, mu * 0 + sigma, value


def dist_transform(mu, logstd, cfg):
    """dist.py:31-47 — returns the (mu, sigma) that parameterise the Normal."""
    if cfg.dist_type == 'normal':
        return mu, torch.exp(logstd)
    if cfg.dist_type == 'dreamerv3_normal':
        lo, hi = cfg.minstd, cfg.maxstd
        return torch.tanh(mu), (hi - lo) * torch.sigmoid(logstd + 2.0) + lo
    raise NotImplementedError(cfg.dist_type)


def normal_log_prob(x, mu, sigma):
    """torch.distributions.Normal.log_prob as used at models.py:106,117."""
    return -((x - mu) ** 2) / (2 * sigma**2) - torch.log(sigma) - LOG_SQRT_2PI


def normal_entropy(sigma):
    """torch.distributions.Normal.entropy as used at models.py:97."""
    return 0.5 + LOG_SQRT_2PI + torch.log(sigma)


def model_act(params, obs, noise, cfg):
    """models.py:110-126 with the sample written as mu + sigma * noise (noise ~ N(0,1) injected)."""
    with torch.no_grad():
        mu, logstd, value = actor_critic_raw(params, obs, cfg)
        mu, sigma = dist_transform(mu, logstd, cfg)
        actions = mu + sigma * noise
        neglogp = -normal_log_prob(actions, mu, sigma).sum(1)
    return {'actions': actions, 'neglogp': neglogp, 'values': value, 'mu': mu, 'sigma': sigma}


def model_forward(params, obs, prev_actions, cfg):
    """models.py:93-108 — training-time forward."""
    mu, logstd, value = actor_critic_raw(params, obs, cfg)
    mu, sigma = dist_transform(mu, logstd, cfg)
    entropy = normal_entropy(sigma).sum(dim=-1)
    prev_neglogp = torch.squeeze(-normal_log_prob(prev_actions, mu, sigma).sum(dim=-1))
    return {'values': value, 'entropy': entropy, 'mu': mu, 'sigma': sigma, 'prev_neglogp': prev_neglogp}


# --------------------------------------------------------------------------
# Losses  (mineral/agents/ppo/ppo.py:449-498)
# --------------------------------------------------------------------------
def smooth_clamp(x, mi, mx):
    """ppo.py:449-450."""
    return 1 / (1 + torch.exp((-(x - mi) / (mx - mi) + 0.5) * 4)) * (mx - mi) + mi


def actor_loss(old_neglogp, neglogp, advantage, e_clip, use_smooth_clamp=False):
    """ppo.py:453-463."""
    clamp = smooth_clamp if use_smooth_clamp else torch.clamp
    ratio = torch.exp(old_neglogp - neglogp)
    surr1 = advantage * ratio
    surr2 = advantage * clamp(ratio, 1.0 - e_clip, 1.0 + e_clip)
    a_loss = torch.max(-surr1, -surr2)
    clip_frac = torch.mean((torch.abs(ratio - 1.0) > e_clip).float()).detach()
    return a_loss, clip_frac


def critic_loss(value_preds, values, e_clip, returns, clip_value_loss):
    """ppo.py:466-476."""
    if clip_value_loss:
        clipped = value_preds + (values - value_preds).clamp(-e_clip, e_clip)
        c_loss = torch.max((values - returns) ** 2, (clipped - returns) ** 2)
    else:
        c_loss = (values - returns) ** 2
    explained_var = (1 - torch.var(returns - values) / (torch.var(returns) + 1e-8)).clamp(0, 1).detach()
    return c_loss, explained_var


def bounds_loss(mu, bounds_type, soft_bound=1.1):
    """ppo.py:479-489."""
    if bounds_type == 'bound':
        hi = torch.clamp_min(mu - soft_bound, 0.0) ** 2
        lo = torch.clamp_max(mu + soft_bound, 0.0) ** 2
        return (lo + hi).sum(dim=-1)
    if bounds_type == 'reg':
        return (mu * mu).sum(dim=-1)
    raise NotImplementedError(bounds_type)


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma):
    """ppo.py:492-498 — p0 = new policy, p1 = stored (old) policy."""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma**2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma**2 + 1e-5))
    return (c1 + c2 - 0.5).sum(dim=-1).mean()


# --------------------------------------------------------------------------
# Rollout buffer  (mineral/agents/ppo/experience.py)
# --------------------------------------------------------------------------
def gae_returns(rewards, values, dones_u8, last_values, gamma, tau):
    """experience.py:96-108.  rewards/values (T,N,1) f32, dones (T,N) u8, last_values (N,1).

    Returns (returns, advs) both (T,N,1)."""
    T = rewards.shape[0]
    advs = torch.zeros_like(rewards)
    returns = torch.zeros_like(rewards)
    last = 0
    for t in reversed(range(T)):
        next_values = last_values if t == T - 1 else values[t + 1]
        nnt = (1.0 - dones_u8.float()[t]).unsqueeze(1)
        delta = rewards[t] + gamma * next_values * nnt - values[t]
        advs[t] = last = delta + gamma * tau * nnt * last
        returns[t] = advs[t] + values[t]
    return returns, advs


def flatten_env_major(x):
    """experience.py:7-14 — (T,N,...) -> (N*T,...), flat index = n*T + t."""
    s = x.shape
    return x.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


def normalize_advantages(returns_flat, values_flat):
    """experience.py:119-120 — whole-batch, unbiased std, +1e-8, squeezed to (B,)."""
    adv = returns_flat - values_flat
    return ((adv - adv.mean()) / (adv.std() + 1e-8)).squeeze(1)


# --------------------------------------------------------------------------
# Optimizer  (torch.optim.Adam / AdamW as constructed at ppo.py:58-59, clip at ppo.py:294-296)
# --------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (norm_type 2): coef = clamp(max_norm / (norm + 1e-6), max=1)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adam_step(params, grads, state, lr, optim_type='AdamW', betas=(0.9, 0.999), eps=1e-8, weight_decay=None):
    """One torch.optim.Adam / AdamW step (single-tensor form, amsgrad False, maximize False).

    `state` = {'step': int, 'exp_avg': {name: t}, 'exp_avg_sq': {name: t}}; updated in place.
    AdamW default weight_decay is 0.01 (decoupled); Adam default is 0 (L2 folded into the gradient).
    """
    if weight_decay is None:
        weight_decay = 0.01 if optim_type == 'AdamW' else 0.0
    b1, b2 = betas
    state['step'] += 1
    step = state['step']
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    for name, g in grads.items():
        p = params[name]
        if optim_type == 'AdamW':
            p.mul_(1 - lr * weight_decay)
        elif weight_decay != 0:
            g = g.add(p, alpha=weight_decay)
        m, v = state['exp_avg'][name], state['exp_avg_sq'][name]
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def adam_state_init(params):
    return {
        'step': 0,
        'exp_avg': {k: torch.zeros_like(v) for k, v in params.items()},
        'exp_avg_sq': {k: torch.zeros_like(v) for k, v in params.items()},
    }


class AdaptiveScheduler:
    """mineral/agents/ppo/utils.py:7-20."""

    def __init__(self, kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2):
        self.kl_threshold, self.min_lr, self.max_lr = kl_threshold, min_lr, max_lr

    def update(self, current_lr, kl_dist):
        lr = current_lr
        if kl_dist > 2.0 * self.kl_threshold:
            lr = max(current_lr / 1.5, self.min_lr)
        if kl_dist < 0.5 * self.kl_threshold:
            lr = min(current_lr * 1.5, self.max_lr)
        return lr


def linear_lr(start_lr, min_lr, max_steps, steps):
    """utils.py:23-32."""
    return max(min_lr, start_lr - start_lr * (steps / float(max_steps)))


def cosine_lr(init_lr, epoch, mini_epochs, agent_steps, max_agent_steps):
    """utils.py:35-37."""
    return init_lr * 0.5 * (1.0 + math.cos(math.pi * (agent_steps + epoch / mini_epochs) / max_agent_steps))


def launcher_seeds(seed):
    """mineral/scripts/run.py:124-127 — five sub-seeds drawn from RandomState(seed)."""
    import numpy as np

    names = ['setup', 'train', 'eval', 'env_train', 'env_eval']
    draws = np.random.RandomState(seed).randint(0, int(1e6), len(names))
    return {k: int(draws[i]) for i, k in enumerate(names)}


# --------------------------------------------------------------------------
# Hyper-parameters and the two hot loops
# --------------------------------------------------------------------------
class PPOHParams:
    """The `agent.ppo.*` keys read at ppo.py:63-93 (defaults = cfgs/agent/DFlexAntPPO.yaml:33-66)."""

    def __init__(self, **kw):
        self.horizon_len = 32
        self.minibatch_size = 2048
        self.mini_epochs = 5
        self.gamma, self.tau = 0.99, 0.95
        self.e_clip = 0.2
        self.use_smooth_clamp = False
        self.clip_value_loss = False
        self.entropy_coef = 0.0
        self.critic_coef = 4.0
        self.bounds_loss_coef = 1e-4
        self.bounds_type = 'bound'
        self.truncate_grads = True
        self.max_grad_norm = 0.5
        self.value_bootstrap = True
        self.normalize_value = True
        self.normalize_input = True
        self.reward_scale = 1.0
        self.optim_type = 'AdamW'
        self.lr = 5e-4
        self.adam_eps = 1e-5
        self.weight_decay = None
        self.lr_schedule = 'kl'
        self.kl_threshold = 0.008
        self.min_lr, self.max_lr = 1e-6, 1e-2
        for k, v in kw.items():
            if not hasattr(self, k):
                raise KeyError(k)
            setattr(self, k, v)


def total_loss(out, mb, hp):
    """ppo.py:277-285 — the scalar that is back-propagated plus the detached diagnostics."""
    a_loss, clip_frac = actor_loss(mb['neglogp'], out['prev_neglogp'], mb['advantages'], hp.e_clip, hp.use_smooth_clamp)
    c_loss, explained_var = critic_loss(mb['values'], out['values'], hp.e_clip, mb['returns'], hp.clip_value_loss)
    b_loss = bounds_loss(out['mu'], hp.bounds_type)
    a_loss, c_loss, entropy, b_loss = [torch.mean(x) for x in (a_loss, c_loss, out['entropy'], b_loss)]
    loss = a_loss + 0.5 * c_loss * hp.critic_coef - entropy * hp.entropy_coef + b_loss * hp.bounds_loss_coef
    diag = {'loss/total': loss.detach(), 'loss/actor': a_loss.detach(), 'loss/critic': c_loss.detach(),
            'loss/bounds': b_loss.detach(), 'loss/entropy': entropy.detach(), 'clip_frac': clip_frac,
            'explained_var': explained_var}
    return loss, diag


def minibatch_step(params, optim_state, obs_rms, mb, cfg, hp, lr, grad_hook=None):
    """One pass of the loop body at ppo.py:254-326 on minibatch `mb` (dict of flat slices).

    Updates params / optim_state in place; returns (new obs_rms, new mu, new sigma, diagnostics).
    `grad_hook(grads: dict) -> dict` lets a multi-rank harness average gradients (the DDP
    all-reduce inside ppo.py:292) before clipping.
    """
    obs = mb['obs']
    if hp.normalize_input:
        obs_rms = rms_update(obs_rms, obs)                      # ppo.py:261
        obs_in = rms_normalize(obs_rms, obs)                    # ppo.py:262
    else:
        obs_in = obs
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    out = model_forward(leaves, obs_in, mb['actions'], cfg)     # ppo.py:270
    loss, diag = total_loss(out, mb, hp)
    grads = torch.autograd.grad(loss, list(leaves.values()))    # ppo.py:291-292
    grads = OrderedDict(zip(leaves.keys(), grads))
    if grad_hook is not None:
        grads = grad_hook(grads)
    if hp.truncate_grads:                                       # ppo.py:294-296
        clipped, gnorm = clip_grad_norm(list(grads.values()), hp.max_grad_norm)
        diag['grad_norm/all'] = gnorm
        clipped = OrderedDict(zip(grads.keys(), clipped))
    else:
        clipped = grads
    adam_step(params, clipped, optim_state, lr, hp.optim_type, eps=hp.adam_eps, weight_decay=hp.weight_decay)  # :300
    mu, sigma = out['mu'].detach(), out['sigma'].detach()
    diag['kl'] = policy_kl(mu, sigma, mb['mu'], mb['sigma'])      # ppo.py:303
    diag['grads'] = grads
    return obs_rms, mu, sigma, diag


def train_epoch(params, optim_state, obs_rms, data, cfg, hp, lr, scheduler=None, grad_hook=None, record=None, optim_lr=None):
    """ppo.py:248-350 — mini_epochs x (B/mb) contiguous minibatches, mu/sigma write-back, KL LR schedule.

    `lr` is the agent's `last_lr`; `optim_lr` is the learning rate the optimizer's param group holds when that differs
    (right after PPO.load, ppo.py:429-446: the optimizer state is restored, `last_lr` is not checkpointed).  The optimizer
    steps with its own value until the end of the first mini-epoch, where ppo.py:344-345 copies `last_lr` into it.

    `data` is the flattened dict of experience.py:110-121 (keys obs, actions, neglogp, values, returns,
    advantages, mu, sigma); `data['mu']`/`data['sigma']` are overwritten in place as at ppo.py:313.
    Returns (obs_rms, lr, results) where results maps diagnostic name -> list of scalars.
    """
    B = data['obs'].shape[0]
    mbs = hp.minibatch_size
    assert B % mbs == 0                                          # ppo.py:83
    results = {}
    step_lr = lr if optim_lr is None else optim_lr
    for mini_ep in range(hp.mini_epochs):
        kls = []
        for i in range(B // mbs):
            sl = slice(i * mbs, (i + 1) * mbs)                    # experience.py:61-64
            mb = {k: v[sl] for k, v in data.items()}
            obs_rms, mu, sigma, diag = minibatch_step(params, optim_state, obs_rms, mb, cfg, hp, step_lr, grad_hook)
            data['mu'][sl] = mu                                   # experience.py:83-87
            data['sigma'][sl] = sigma
            kls.append(diag['kl'])
            for k, v in diag.items():
                if k != 'grads':
                    results.setdefault(k, []).append(float(v))
            if record is not None:
                record(mini_ep, i, diag)
        avg_kl = torch.mean(torch.stack(kls))                     # ppo.py:334
        results.setdefault('avg_kl', []).append(float(avg_kl))
        if hp.lr_schedule == 'kl':                                # ppo.py:337-338
            lr = scheduler.update(lr, avg_kl.item())
        step_lr = lr                                              # ppo.py:344-345
        results.setdefault('lr', []).append(lr)
    return obs_rms, lr, results


def play_steps(params, obs_rms, value_rms, env, obs, noise, cfg, hp):
    """ppo.py:133-183 — T policy/env steps into a (T,N,.) buffer, GAE, flatten, value normalisation.

    `env.step(actions) -> (obs, reward (N,), done (N,) bool, infos)`; `noise` is (T,N,A) standard normal.
    Returns (data dict, new value_rms, last obs).
    """
    T, N, A = hp.horizon_len, obs.shape[0], cfg.act_dim
    st = {
        'obs': torch.zeros(T, N, cfg.obs_dim), 'actions': torch.zeros(T, N, A), 'mu': torch.zeros(T, N, A),
        'sigma': torch.zeros(T, N, A), 'neglogp': torch.zeros(T, N), 'values': torch.zeros(T, N, 1),
        'rewards': torch.zeros(T, N, 1), 'dones': torch.zeros(T, N, dtype=torch.uint8),
    }

    def act(o, eps):
        o_in = rms_normalize(obs_rms, o) if hp.normalize_input else o     # ppo.py:126-127
        out = model_act(params, o_in, eps, cfg)
        if hp.normalize_value:
            out['values'] = rms_unnormalize(value_rms, out['values'])     # ppo.py:129-130
        return out

    for t in range(T):
        out = act(obs, noise[t])
        st['obs'][t] = obs                                                # ppo.py:146-148
        for k in ('actions', 'neglogp', 'values', 'mu', 'sigma'):
            st[k][t] = out[k]
        obs, r, dones, infos = env.step(torch.clamp(out['actions'], -1.0, 1.0))   # ppo.py:150-151
        st['dones'][t] = dones                                            # ppo.py:157 (bool -> u8)
        shaped = r.reshape(-1, 1) * hp.reward_scale                        # reward_shaper.py:10-12
        if hp.value_bootstrap and 'time_outs' in infos:                   # ppo.py:159-162
            shaped = shaped + hp.gamma * out['values'] * infos['time_outs'].reshape(-1, 1).float()
        st['rewards'][t] = shaped
    last_values = act(obs, torch.zeros(N, A))['values']                   # ppo.py:169-170
    returns, _ = gae_returns(st['rewards'], st['values'], st['dones'], last_values, hp.gamma, hp.tau)
    data = {k: flatten_env_major(v) for k, v in st.items()}
    data['returns'] = flatten_env_major(returns)
    data['advantages'] = normalize_advantages(data['returns'], data['values'])
    if hp.normalize_value:                                                # ppo.py:177-181
        value_rms = rms_update(value_rms, data['values'])
        data['values'] = rms_normalize(value_rms, data['values'])
        value_rms = rms_update(value_rms, data['returns'])
        data['returns'] = rms_normalize(value_rms, data['returns'])
    return data, value_rms, obs
