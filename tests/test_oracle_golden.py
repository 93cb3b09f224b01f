"""Pins the CPU oracle (`oracle/ppo_oracle.py`) against vectors produced by the reference itself.

The fixtures under `tests/golden/*.npz` were written by `tests/golden/make_golden.py`, which imports the
reference's Python and records its outputs on hash-derived inputs; here the same inputs are rebuilt and
pushed through the oracle.  Index/byte results must be bit-exact; fp32 results agree to a few ulp because
both sides run the same torch CPU kernels.
"""

import numpy as np
import pytest
import torch

import synth
from oracle import ppo_oracle as O

torch.set_num_threads(1)

NETS = {
    'small_dv3': dict(obs=5, act=3, units=[16, 8], norm='LayerNorm', act_type='SiLU', fixed_sigma=False,
                      dist_type='dreamerv3_normal'),
    'full_dv3': dict(obs=64, act=12, units=[512, 256, 128], norm='LayerNorm', act_type='SiLU', fixed_sigma=False,
                     dist_type='dreamerv3_normal'),
    'small_normal': dict(obs=5, act=3, units=[16, 8], norm=None, act_type='ELU', fixed_sigma=True,
                         dist_type='normal'),
    'mid_dv3': dict(obs=64, act=12, units=[128, 64, 32], norm='LayerNorm', act_type='SiLU', fixed_sigma=False,
                    dist_type='dreamerv3_normal'),
    'cartpole': dict(obs=4, act=1, units=[32, 32], norm=None, act_type='ELU', fixed_sigma=True, dist_type='normal'),
    # LayerNorm + ELU towers (mlp_kwargs of DFlexAntSHAC.yaml / DFlexAntBPTT.yaml) in the PPO ActorCritic
    'mid_lnelu': dict(obs=64, act=12, units=[128, 64, 32], norm='LayerNorm', act_type='ELU', fixed_sigma=False,
                      dist_type='dreamerv3_normal'),
}


def net_cfg(spec):
    return O.NetConfig(spec['obs'], spec['act'], spec['units'], spec['norm'], spec['act_type'], spec['fixed_sigma'],
                       spec['dist_type'], 0.1, 1.0)


def net_params(spec, seed):
    shapes = synth.actor_critic_shapes(spec['obs'], spec['act'], spec['units'], spec['norm'] is not None, True,
                                       spec['fixed_sigma'])
    return synth.make_params(shapes, seed)


def close(a, b, rtol=1e-6, atol=1e-7):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def exact(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.dtype == b.dtype and a.shape == b.shape
    assert np.array_equal(a, b)


@pytest.mark.parametrize('tag,T,N,Odim,A', [('small', 4, 8, 5, 3), ('mid', 32, 64, 64, 12)])
def test_gae_flatten_advnorm(golden, tag, T, N, Odim, A):
    g = golden(f'g1_gae_{tag}')
    rewards, values = synth.gaussianish((T, N, 1), 11), synth.gaussianish((T, N, 1), 12)
    dones = (synth.lcg_t((T, N), 14) > 0.4).to(torch.uint8)
    last_values = synth.gaussianish((N, 1), 18)
    returns, _ = O.gae_returns(rewards, values, dones, last_values, 0.99, 0.95)
    exact(returns, g['returns'])
    exact(O.flatten_env_major(returns), g['flat_returns'])
    exact(O.flatten_env_major(values), g['flat_values'])
    exact(O.flatten_env_major(dones), g['flat_dones'])
    # index map: payload is arange, so this is the bit-exact check of flat = n*T + t
    obs = torch.arange(T * N * Odim, dtype=torch.float32).reshape(T, N, Odim)
    exact(O.flatten_env_major(obs), g['flat_obs'])
    exact(O.normalize_advantages(O.flatten_env_major(returns), O.flatten_env_major(values)), g['advantages'])
    # contiguous minibatch slice 1 of 2 and the mu/sigma write-back
    g2 = golden(f'g2_slice_{tag}')
    mb = T * N // 2
    exact(O.flatten_env_major(values)[mb:2 * mb], g2['mb_values'])
    exact(O.flatten_env_major(obs)[mb:2 * mb], g2['mb_obs'])
    mu = O.flatten_env_major(synth.gaussianish((T, N, A), 16)).clone()
    mu[mb:2 * mb] = 7.0
    exact(mu, g2['mu_after'])


@pytest.mark.parametrize('tag,kw', [
    ('ppo', dict(eps=1e-5, clamp=(-5.0, 5.0), initial_count=1.0, dtype=torch.float64, correction=1)),
    ('sac', dict(eps=1e-4, clamp=None, initial_count=1e-4, dtype=torch.float32, correction=1)),
    ('shac', dict(eps=1e-5, clamp=None, initial_count=1e-4, dtype=torch.float32, correction=0)),
])
def test_running_mean_std(golden, tag, kw):
    g = golden(f'g3_rms_{tag}')
    st = O.rms_init((6,), kw['initial_count'], kw['dtype'])
    for i in range(5):
        x = synth.gaussianish((37 + 11 * i, 6), 100 + i) * (1.0 + i) + 0.3 * i
        st = O.rms_update(st, x, kw['correction'])
        exact(st['running_mean'], g[f'mean{i}'])
        exact(st['running_var'], g[f'var{i}'])
        exact(st['running_count'], g[f'count{i}'])
        exact(O.rms_normalize(st, x * 3, kw['eps'], kw['clamp']), g[f'norm{i}'])
        exact(O.rms_unnormalize(st, x * 3, kw['eps'], kw['clamp']), g[f'unnorm{i}'])


@pytest.mark.parametrize('tag', ['small_dv3', 'full_dv3', 'small_normal', 'mid_lnelu'])
def test_actor_critic_forward(golden, tag):
    g = golden(f'g4_model_{tag}')
    spec = NETS[tag]
    cfg, params = net_cfg(spec), net_params(spec, 3)
    obs = synth.gaussianish((16, spec['obs']), 41)
    prev = synth.gaussianish((16, spec['act']), 42)
    noise = synth.gaussianish((16, spec['act']), 43)
    out = O.model_forward(params, obs, prev, cfg)
    for k in ('values', 'entropy', 'mu', 'sigma', 'prev_neglogp'):
        close(out[k], g[k])
    act = O.model_act(params, obs, noise, cfg)
    for k in ('actions', 'neglogp', 'values', 'mu', 'sigma'):
        close(act[k], g[f'act_{k}'])


def test_losses(golden):
    g = golden('g5_losses')
    B, A = 64, 4
    old = synth.gaussianish((B,), 51)
    new = old + synth.gaussianish((B,), 52) * 0.3
    adv = synth.gaussianish((B,), 53)
    vp, v, ret = synth.gaussianish((B, 1), 54), synth.gaussianish((B, 1), 55), synth.gaussianish((B, 1), 56)
    mu = synth.gaussianish((B, A), 57) * 1.2
    s0, s1 = synth.lcg_t((B, A), 58) + 0.7, synth.lcg_t((B, A), 59) + 0.7
    mu1 = mu + synth.gaussianish((B, A), 60) * 0.1
    a_hard, cf = O.actor_loss(old, new, adv, 0.2, False)
    exact(a_hard, g['a_hard'])
    exact(cf, g['clip_frac'])
    exact(O.actor_loss(old, new, adv, 0.2, True)[0], g['a_soft'])
    c0, ev = O.critic_loss(vp, v, 0.2, ret, False)
    exact(c0, g['c_plain'])
    exact(ev, g['explained_var'])
    exact(O.critic_loss(vp, v, 0.2, ret, True)[0], g['c_clip'])
    exact(O.bounds_loss(mu, 'bound'), g['b_bound'])
    exact(O.bounds_loss(mu, 'reg'), g['b_reg'])
    exact(O.policy_kl(mu, s0, mu1, s1), g['kl'])
    exact(O.smooth_clamp(synth.gaussianish((B,), 61) + 1, 0.8, 1.2), g['smooth'])


def test_seeds_and_schedulers(golden):
    g = golden('g9_g10_seeds_sched')
    for s in (0, 42, 100):
        assert list(O.launcher_seeds(s).values()) == list(g[f'seed{s}'])
    assert O.launcher_seeds(42) == {'setup': 121958, 'train': 671155, 'eval': 131932, 'env_train': 365838,
                                    'env_eval': 259178}
    sch = O.AdaptiveScheduler(0.008, 1e-6, 1e-2)
    kls = synth.lcg((40,), 77, 0.04, np.float64) + 0.02
    lr, lrs = 5e-4, []
    for k in kls:
        lr = sch.update(lr, float(k))
        lrs.append(lr)
    assert np.array_equal(np.array(lrs), g['kl_lrs'])
    lin = [O.linear_lr(5e-4, 1e-6, 1000000, s) for s in range(0, 1200000, 100000)]
    assert np.array_equal(np.array(lin), g['lin_lrs'])
    cos = [O.cosine_lr(5e-4, e, 5, s, 4100000) for e in range(5) for s in (0, 1000000, 4000000)]
    assert np.array_equal(np.array(cos), g['cos_lrs'])


TRAIN_CASES = [
    ('g8_train_small_dv3_adamw', 'small_dv3', dict(N=16, T=8, mb=32, mini_epochs=3, epochs=3, optim='AdamW')),
    ('g8_train_small_dv3_adam', 'small_dv3', dict(N=16, T=8, mb=64, mini_epochs=2, epochs=2, optim='Adam')),
    ('g8_train_small_normal_adam', 'small_normal', dict(N=16, T=8, mb=64, mini_epochs=2, epochs=2, optim='Adam')),
    ('g8_train_mid_dv3_adamw', 'mid_dv3', dict(N=64, T=8, mb=128, mini_epochs=2, epochs=2, optim='AdamW')),
    ('g8_train_cartpole_adam', 'cartpole', dict(N=64, T=8, mb=128, mini_epochs=2, epochs=2, optim='Adam')),
    # the DFlexAntPPO network itself
    ('g8_train_full_dv3_adamw', 'full_dv3', dict(N=64, T=8, mb=256, mini_epochs=2, epochs=2, optim='AdamW')),
    ('g8_train_mid_lnelu_adamw', 'mid_lnelu', dict(N=64, T=8, mb=128, mini_epochs=2, epochs=2, optim='AdamW')),
]


def run_oracle_training(spec, c, on_epoch=None, first_step=None, resume=None):
    """The reference's PPO.train() loop (ppo.py:185-199) driven through the oracle on GoldenEnv.  `resume`: a checkpoint
    dict in the reference's format (ppo.py:416-426) loaded as PPO.load does before training (ppo.py:429-446)."""
    cfg, params = net_cfg(spec), net_params(spec, 5)
    hp = O.PPOHParams(horizon_len=c['T'], minibatch_size=c['mb'], mini_epochs=c['mini_epochs'],
                      optim_type=c['optim'], reward_scale=0.5)
    env = synth.GoldenEnv(c['N'], spec['obs'], spec['act'])
    obs = env.reset()
    obs_rms, value_rms = O.rms_init((spec['obs'],)), O.rms_init((1,))
    optim_state = O.adam_state_init(params)
    sched = O.AdaptiveScheduler(hp.kl_threshold, hp.min_lr, hp.max_lr)
    lr = hp.lr
    step = 0
    first_epoch, optim_lr = 0, None
    if resume is not None:
        names = list(params)
        for n in names:
            params[n] = resume['model'][n].clone()
        for i, n in enumerate(names):
            st = resume['optim']['state'][i]
            optim_state['exp_avg'][n], optim_state['exp_avg_sq'][n] = st['exp_avg'].clone(), st['exp_avg_sq'].clone()
            optim_state['step'] = int(st['step'])
        obs_rms = {k.split('.', 1)[1]: v.clone() for k, v in resume['obs_rms'].items()}
        value_rms = {k: v.clone() for k, v in resume['value_rms'].items()}
        optim_lr = float(resume['optim']['param_groups'][0]['lr'])   # last_lr itself is not in the checkpoint
        first_epoch = int(resume['epoch']) + 1
    for e in range(first_epoch, c['epochs']):
        noise = torch.stack([synth.gaussianish((c['N'], spec['act']), 900 + step + t) for t in range(c['T'])])
        step += c['T'] + 1          # the bootstrap forward at ppo.py:169 also draws (and discards) a sample
        data, value_rms, obs = O.play_steps(params, obs_rms, value_rms, env, obs, noise, cfg, hp)
        pre = {k: v.clone() for k, v in data.items()}
        rec = None
        if e == 0 and first_step is not None:
            def rec(mini_ep, i, diag, _seen=[False]):
                if not _seen[0]:
                    _seen[0] = True
                    first_step(diag, {k: v.clone() for k, v in params.items()}, optim_state)
        obs_rms, lr, res = O.train_epoch(params, optim_state, obs_rms, data, cfg, hp, lr, sched, record=rec, optim_lr=optim_lr)
        optim_lr = None
        if on_epoch is not None:
            on_epoch(e, pre, data, res, lr, obs_rms, value_rms, params)
    return params


@pytest.mark.parametrize('fixture,net,c', TRAIN_CASES)
def test_end_to_end_training(golden, fixture, net, c):
    g = golden(fixture)
    spec = NETS[net]
    tol = dict(rtol=2e-5, atol=2e-6)

    def first_step(diag, params_after, optim_state):
        if 's0_g_mu.weight' not in g:
            return
        # gradients seen by optimizer.step are post-clip (ppo.py:296-300)
        grads = diag['grads']
        clipped, _ = O.clip_grad_norm(list(grads.values()), 0.5)
        for (n, _), gc in zip(grads.items(), clipped):
            close(gc, g[f's0_g_{n}'], rtol=1e-4, atol=1e-7)
            close(params_after[n], g[f's0_p_{n}'], **tol)
            close(optim_state['exp_avg'][n], g[f's0_m_{n}'], rtol=1e-4, atol=1e-8)
            close(optim_state['exp_avg_sq'][n], g[f's0_v_{n}'], rtol=1e-4, atol=1e-12)

    def on_epoch(e, pre, data, res, lr, obs_rms, value_rms, params):
        if e == 0:
            exact(pre['dones'], g['e0_pre_dones'])
            for k in ('values', 'returns', 'advantages', 'neglogp', 'actions', 'mu', 'sigma', 'rewards', 'obs'):
                close(pre[k], g[f'e0_pre_{k}'], **tol)
        n_mb = c['N'] * c['T'] // c['mb']
        kl = np.array(res['kl']).reshape(c['mini_epochs'], n_mb).mean(1)
        close(kl, g[f'e{e}_kl'], rtol=1e-3, atol=1e-7)
        assert lr == float(g[f'e{e}_last_lr'])
        for k in ('loss/total', 'loss/actor', 'loss/critic', 'clip_frac', 'explained_var', 'grad_norm/all'):
            close(np.array(res[k], dtype=np.float32), g[f'e{e}_{k.replace("/", "_")}'], rtol=2e-4, atol=2e-6)
        close(data['mu'], g[f'e{e}_data_mu'], **tol)
        close(data['sigma'], g[f'e{e}_data_sigma'], **tol)
        # the batch moments are fp32 torch reductions whose association depends on the layout of the minibatch view: one
        # fp32 ulp of a batch mean is 6e-8 relative
        # (ELU towers: the parameters after the first update differ from the recorded ones in the last fp32 bit -- the
        # LayerNorm+SiLU runs are bit-equal --, so the later rollouts, whose observations depend on the actions, differ at
        # fp32 rounding level: 2e-8 absolute on statistics of O(1) observations)
        rms_atol = 2e-8 if spec['act_type'] == 'ELU' else 1e-9
        close(obs_rms['running_mean'], g[f'e{e}_obs_rms_mean'], rtol=1e-7, atol=rms_atol)
        close(obs_rms['running_var'], g[f'e{e}_obs_rms_var'], rtol=1e-7, atol=rms_atol)
        exact(obs_rms['running_count'], g[f'e{e}_obs_rms_count'])
        close(value_rms['running_mean'], g[f'e{e}_value_rms_mean'], rtol=1e-6, atol=1e-9)
        close(value_rms['running_var'], g[f'e{e}_value_rms_var'], rtol=1e-6, atol=1e-9)
        if f'e{e}_p_mu.weight' in g:
            for n, p in params.items():
                close(p, g[f'e{e}_p_{n}'], **tol)

    run_oracle_training(spec, c, on_epoch, first_step)
    assert int(g['epochs_run']) == c['epochs']
    assert int(g['agent_steps']) == c['epochs'] * c['N'] * c['T']


def test_training_continues_from_a_checkpoint_written_by_the_reference(golden):
    """tests/golden/g16_reference_checkpoint.pth was written by the reference's own PPO.save after one epoch; g16_ckpt_continued
    holds what the reference did after PPO.load of it into a fresh agent (one more epoch).  The oracle resumes the same way:
    parameters, both Adam moments and the step count, both running statistics, and the optimizer's learning rate (the
    agent's last_lr is not checkpointed and restarts from the configured value)."""
    import os
    ckpt = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'g16_reference_checkpoint.pth'), map_location='cpu', weights_only=True)
    assert set(ckpt) == {'epoch', 'mini_epoch', 'agent_steps', 'model', 'optim', 'obs_rms', 'value_rms'}
    g = golden('g16_ckpt_continued')
    spec, c = NETS['mid_dv3'], dict(N=64, T=8, mb=128, mini_epochs=2, epochs=2, optim='AdamW')
    seen = {}

    def on_epoch(e, pre, data, res, lr, obs_rms, value_rms, params):
        seen['e'] = e
        n_mb = c['N'] * c['T'] // c['mb']
        close(np.array(res['kl']).reshape(c['mini_epochs'], n_mb).mean(1), g[f'e{e}_kl'], rtol=1e-3, atol=1e-7)
        assert lr == float(g[f'e{e}_last_lr'])
        exact(obs_rms['running_count'], g[f'e{e}_obs_rms_count'])
        close(obs_rms['running_mean'], g[f'e{e}_obs_rms_mean'], rtol=1e-7, atol=1e-9)
        close(value_rms['running_var'], g[f'e{e}_value_rms_var'], rtol=1e-6, atol=1e-9)
        for n, p in params.items():
            close(p, g[f'e{e}_p_{n}'], rtol=2e-5, atol=2e-6)

    run_oracle_training(spec, c, on_epoch, resume=ckpt)
    assert seen['e'] == 1 and int(g['epochs_run']) == 2
