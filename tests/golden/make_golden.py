"""Generate golden vectors by running the REFERENCE implementation (read-only checkout) on CPU.

Run once, in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [/root/reference]

The reference package needs `omegaconf`, `wandb` and `torch.utils.tensorboard`, none of which are
installed here; three in-process stub modules stand in for them (they are never on the numeric path).
Inputs come from `synth.py` (hash-derived, rebuilt exactly by the tests), so the `.npz` files written
next to this script hold OUTPUTS only.  Nothing of the reference's source is stored.
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
sys.path.insert(0, REF)
sys.dont_write_bytecode = True


# ---- stubs (must exist before `import mineral.agents`) -------------------------------------------
class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in d.items()})
        return d


def _plain(d):
    return {k: _plain(v) for k, v in d.items()} if isinstance(d, dict) else d


omegaconf = types.ModuleType('omegaconf')


class _OmegaConf:
    @staticmethod
    def to_container(cfg, resolve=True, **kw):
        return _plain(cfg)

    @staticmethod
    def register_new_resolver(*a, **k):
        pass


omegaconf.OmegaConf = _OmegaConf
omegaconf.DictConfig = AttrDict
sys.modules['omegaconf'] = omegaconf

wandb = types.ModuleType('wandb')
wandb.log = lambda *a, **k: None
wandb.Histogram = lambda v: v
sys.modules['wandb'] = wandb

import torch  # noqa: E402

tb = types.ModuleType('torch.utils.tensorboard')


class _SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None


tb.SummaryWriter = _SummaryWriter
sys.modules['torch.utils.tensorboard'] = tb

torch.set_num_threads(1)

import synth  # noqa: E402
from mineral.agents.ppo import ppo as ref_ppo  # noqa: E402
from mineral.agents.ppo.experience import ExperienceBuffer  # noqa: E402
from mineral.agents.ppo.models import ActorCritic  # noqa: E402
from mineral.agents.ppo.utils import AdaptiveScheduler, LinearScheduler, adjust_learning_rate_cos  # noqa: E402
from mineral.common.normalizers import RunningMeanStd  # noqa: E402

OUT = {}


def save(name, **arrays):
    arrs = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        arrs[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, keys={len(arrs)}')


# ---- G1 / G2: GAE + prepare_training --------------------------------------------------------------
def g1_g2():
    for tag, T, N, O, A in (('small', 4, 8, 5, 3), ('mid', 32, 64, 64, 12)):
        buf = ExperienceBuffer(N, T, T * N, T * N // 2, {'obs': (O,)}, A, 'cpu', __import__('re').compile('$^'))
        s = buf.storage_dict
        s['rewards'][:] = synth.gaussianish((T, N, 1), 11)
        s['values'][:] = synth.gaussianish((T, N, 1), 12)
        s['neglogp'][:] = synth.gaussianish((T, N), 13)
        s['dones'][:] = (synth.lcg_t((T, N), 14) > 0.4).to(torch.uint8)
        s['actions'][:] = synth.gaussianish((T, N, A), 15)
        s['mu'][:] = synth.gaussianish((T, N, A), 16)
        s['sigma'][:] = synth.lcg_t((T, N, A), 17) + 0.6
        s['obses']['obs'][:] = torch.arange(T * N * O, dtype=torch.float32).reshape(T, N, O)  # index payload
        last_values = synth.gaussianish((N, 1), 18)
        buf.compute_return(last_values, 0.99, 0.95)
        returns_tn = s['returns'].clone()
        d = buf.prepare_training()
        save(f'g1_gae_{tag}', returns=returns_tn, flat_returns=d['returns'], flat_values=d['values'],
             advantages=d['advantages'], flat_obs=d['obses']['obs'], flat_neglogp=d['neglogp'],
             flat_dones=d['dones'], flat_actions=d['actions'])
        # minibatch slicing + mu/sigma write-back (experience.py:61-87)
        item = buf[1]
        buf.update_mu_sigma(item[3] * 0 + 7.0, item[4] * 0 + 9.0)
        save(f'g2_slice_{tag}', mb_values=item[0], mb_neglogp=item[1], mb_adv=item[2], mb_obs=item[7]['obs'],
             mu_after=buf.data_dict['mu'], sigma_after=buf.data_dict['sigma'])


# ---- G3: RunningMeanStd -----------------------------------------------------------------------------
def g3():
    variants = {
        'ppo': dict(eps=1e-5, with_clamp=True, initial_count=1, dtype=torch.float64),
        'sac': dict(eps=1e-4, initial_count='eps', dtype=torch.float32),
        'shac': dict(eps=1e-5, correction=0, initial_count=1e-4, dtype=torch.float32),
    }
    for tag, kw in variants.items():
        rms = RunningMeanStd((6,), **kw)
        out = {}
        for i in range(5):
            x = synth.gaussianish((37 + 11 * i, 6), 100 + i) * (1.0 + i) + 0.3 * i
            rms.update(x)
            out[f'mean{i}'], out[f'var{i}'], out[f'count{i}'] = rms.running_mean, rms.running_var, rms.running_count
            out[f'norm{i}'] = rms.normalize(x * 3)
            out[f'unnorm{i}'] = rms.unnormalize(x * 3)
        save(f'g3_rms_{tag}', **out)


# ---- G4: ActorCritic forward / act ------------------------------------------------------------------
NETS = {
    'small_dv3': dict(obs=5, act=3, units=[16, 8], norm='LayerNorm', act_type='SiLU', fixed_sigma=False,
                      dist=dict(dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0)),
    'full_dv3': dict(obs=64, act=12, units=[512, 256, 128], norm='LayerNorm', act_type='SiLU', fixed_sigma=False,
                     dist=dict(dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0)),
    'small_normal': dict(obs=5, act=3, units=[16, 8], norm=None, act_type='ELU', fixed_sigma=True,
                         dist=dict(dist_type='normal')),
    # the network of cfgs/agent/CartpolePPO.yaml (no normalisation layer, ELU, state-independent log-std) at widths the
    # device kernels take
    'cartpole': dict(obs=4, act=1, units=[32, 32], norm=None, act_type='ELU', fixed_sigma=True, dist=dict(dist_type='normal')),
}
# LayerNorm + ELU towers (the mlp_kwargs of DFlexAntSHAC.yaml / DFlexAntBPTT.yaml) in the PPO ActorCritic; kept out of NETS so
# that g4() keeps writing exactly the files it wrote before
LNELU = dict(obs=64, act=12, units=[128, 64, 32], norm='LayerNorm', act_type='ELU', fixed_sigma=False,
             dist=dict(dist_type='dreamerv3_normal', minstd=0.1, maxstd=1.0))


def build_model(spec, seed):
    m = ActorCritic({'obs': (spec['obs'],)}, spec['act'],
                    mlp_kwargs=dict(units=spec['units'], norm_type=spec['norm'], act_type=spec['act_type']),
                    fixed_sigma=spec['fixed_sigma'], actor_dist_kwargs=spec['dist'])
    shapes = synth.actor_critic_shapes(spec['obs'], spec['act'], spec['units'], spec['norm'] is not None, True,
                                       spec['fixed_sigma'])
    names = [n for n, _ in m.named_parameters()]
    assert names == [n for n, _ in shapes], (names, shapes)
    params = synth.make_params(shapes, seed)
    m.load_state_dict(params, strict=True)
    return m


def g4(nets=None):
    for tag, spec in (nets or NETS).items():
        m = build_model(spec, 3)
        B = 16
        obs = synth.gaussianish((B, spec['obs']), 41)
        prev = synth.gaussianish((B, spec['act']), 42)
        out = m({'obs': obs, 'prev_actions': prev})
        noise = synth.gaussianish((B, spec['act']), 43)
        orig = torch.distributions.Normal.sample
        torch.distributions.Normal.sample = lambda self, sample_shape=torch.Size(): self.loc + self.scale * noise
        try:
            act = m.act({'obs': obs})
        finally:
            torch.distributions.Normal.sample = orig
        save(f'g4_model_{tag}', values=out['values'], entropy=out['entropy'], mu=out['mu'], sigma=out['sigma'],
             prev_neglogp=out['prev_neglogp'], act_actions=act['actions'], act_neglogp=act['neglogp'],
             act_values=act['values'], act_mu=act['mu'], act_sigma=act['sigma'])


# ---- G5: losses -------------------------------------------------------------------------------------
def g5():
    B, A = 64, 4
    old = synth.gaussianish((B,), 51)
    new = old + synth.gaussianish((B,), 52) * 0.3
    adv = synth.gaussianish((B,), 53)
    vp, v, ret = synth.gaussianish((B, 1), 54), synth.gaussianish((B, 1), 55), synth.gaussianish((B, 1), 56)
    mu = synth.gaussianish((B, A), 57) * 1.2
    s0, s1 = synth.lcg_t((B, A), 58) + 0.7, synth.lcg_t((B, A), 59) + 0.7
    mu1 = mu + synth.gaussianish((B, A), 60) * 0.1
    a_hard, cf = ref_ppo.actor_loss(old, new, adv, 0.2, False)
    a_soft, _ = ref_ppo.actor_loss(old, new, adv, 0.2, True)
    c0, ev = ref_ppo.critic_loss(vp, v, 0.2, ret, False)
    c1, _ = ref_ppo.critic_loss(vp, v, 0.2, ret, True)
    save('g5_losses', a_hard=a_hard, clip_frac=cf, a_soft=a_soft, c_plain=c0, c_clip=c1, explained_var=ev,
         b_bound=ref_ppo.bounds_loss(mu, 'bound'), b_reg=ref_ppo.bounds_loss(mu, 'reg'),
         kl=ref_ppo.policy_kl(mu, s0, mu1, s1), smooth=ref_ppo.smooth_clamp(synth.gaussianish((B,), 61) + 1, 0.8, 1.2))


# ---- G6..G8: the real agent on a deterministic env ---------------------------------------------------
def make_cfg(spec, N, T, mb, mini_epochs, epochs, optim_type, logdir, hp_over=None):
    ppo = dict(
        multi_gpu=False, num_actors=N, reward_shaper=dict(fn='scale', scale=0.5), max_agent_steps=epochs * N * T,
        horizon_len=T, minibatch_size=mb, normalize_value=True, value_bootstrap=True, clip_value_loss=False,
        normalize_advantage=True, gamma=0.99, tau=0.95, entropy_coef=0.0, e_clip=0.2, use_smooth_clamp=False,
        critic_coef=4, bounds_loss_coef=0.0001, bounds_type='bound', mini_epochs=mini_epochs, optim_type=optim_type,
        optim_kwargs=dict(lr=5e-4, eps=1e-5), lr_schedule='kl', kl_threshold=0.008, max_grad_norm=0.5,
        truncate_grads=True)
    ppo.update(hp_over or {})
    cfg = dict(
        rl_device='cpu', multi_gpu=False, env_render=False, run='train', seed=42,
        task=dict(name='Golden', env=dict(numEnvs=N), env_autoresets=True),
        agent=dict(
            algo='PPO', print_every=-1, ckpt_every=-1, tracker_len=100, metrics_kwargs={},
            network=dict(
                normalize_input=True, actor_critic='ActorCritic',
                actor_critic_kwargs=dict(
                    separate_value_mlp=True,
                    mlp_kwargs=dict(units=spec['units'], norm_type=spec['norm'], act_type=spec['act_type']),
                    fixed_sigma=spec['fixed_sigma'], actor_dist_kwargs=spec['dist'])),
            ppo=ppo))
    return AttrDict.wrap(cfg)


def run_agent(tag, spec, N, T, mb, mini_epochs, epochs, optim_type, hp_over=None, dump_first_step=False,
              params_every_epoch=True, save_ckpt=None, load_ckpt=None):
    """save_ckpt: after training, write the REFERENCE agent's own checkpoint (PPO.save, ppo.py:416-426) to that path.
    load_ckpt: load that checkpoint into a fresh reference agent (PPO.load, ppo.py:429-446) before training, as
    `run=train ckpt=...` does (scripts/run.py:146-148); `epochs` then counts the total including the loaded ones."""
    logdir = f'/tmp/golden_{tag}'
    cfg = make_cfg(spec, N, T, mb, mini_epochs, epochs, optim_type, logdir, hp_over)
    env = synth.GoldenEnv(N, spec['obs'], spec['act'])
    agent = ref_ppo.PPO(cfg, logdir=logdir, accelerator=None, datasets=None, env=env)
    real_save = agent.save
    agent.save = lambda *a, **k: None
    agent._checkpoint_save = lambda *a, **k: None
    shapes = synth.actor_critic_shapes(spec['obs'], spec['act'], spec['units'], spec['norm'] is not None, True,
                                       spec['fixed_sigma'])
    agent.model.load_state_dict(synth.make_params(shapes, 5), strict=True)
    if load_ckpt is not None:
        agent.load(load_ckpt, ckpt_keys='')

    step_counter = [0]
    orig_sample = torch.distributions.Normal.sample

    def fake_sample(self, sample_shape=torch.Size()):
        eps = synth.gaussianish(tuple(self.loc.shape), 900 + step_counter[0])
        step_counter[0] += 1
        return self.loc + self.scale * eps

    out = {}
    orig_train_epoch = agent.train_epoch  # already timer-wrapped

    def spy_train_epoch():
        e = agent.epoch
        d = agent.storage.data_dict
        if e == 0 and load_ckpt is None:
            for k in ('values', 'returns', 'advantages', 'neglogp', 'actions', 'mu', 'sigma', 'rewards', 'dones'):
                out[f'e0_pre_{k}'] = d[k].clone()
            out['e0_pre_obs'] = d['obses']['obs'].clone()
            out['e0_value_rms_mean'] = agent.value_rms.running_mean.clone()
            out['e0_value_rms_var'] = agent.value_rms.running_var.clone()
            out['e0_value_rms_count'] = agent.value_rms.running_count.clone()
        res = orig_train_epoch()
        out[f'e{e}_kl'] = torch.stack(res['avg_kl'])
        for k in ('loss/total', 'loss/actor', 'loss/critic', 'loss/bounds', 'loss/entropy', 'clip_frac',
                  'explained_var', 'grad_norm/all'):
            out[f'e{e}_{k.replace("/", "_")}'] = torch.stack(res[k])
        out[f'e{e}_last_lr'] = np.float64(agent.last_lr)
        out[f'e{e}_data_mu'] = agent.storage.data_dict['mu'].clone()
        out[f'e{e}_data_sigma'] = agent.storage.data_dict['sigma'].clone()
        out[f'e{e}_obs_rms_mean'] = agent.obs_rms['obs'].running_mean.clone()
        out[f'e{e}_obs_rms_var'] = agent.obs_rms['obs'].running_var.clone()
        out[f'e{e}_obs_rms_count'] = agent.obs_rms['obs'].running_count.clone()
        out[f'e{e}_value_rms_mean'] = agent.value_rms.running_mean.clone()
        out[f'e{e}_value_rms_var'] = agent.value_rms.running_var.clone()
        if params_every_epoch or e == epochs - 1:
            for n, p in agent.model.named_parameters():
                out[f'e{e}_p_{n}'] = p.detach().clone()
        return res

    agent.train_epoch = spy_train_epoch

    if dump_first_step:
        # capture raw gradients + optimizer state of the very first optimizer step
        orig_step = agent.optim.step
        seen = [False]

        def spy_step(*a, **k):
            if not seen[0]:
                for n, p in agent.model.named_parameters():
                    out[f's0_g_{n}'] = p.grad.detach().clone()   # already clipped (clip precedes step)
            r = orig_step(*a, **k)
            if not seen[0]:
                seen[0] = True
                for n, p in agent.model.named_parameters():
                    st = agent.optim.state[p]
                    out[f's0_p_{n}'] = p.detach().clone()
                    out[f's0_m_{n}'] = st['exp_avg'].clone()
                    out[f's0_v_{n}'] = st['exp_avg_sq'].clone()
            return r

        agent.optim.step = spy_step

    torch.distributions.Normal.sample = fake_sample
    try:
        agent.train()
    finally:
        torch.distributions.Normal.sample = orig_sample
    out['epochs_run'] = np.int64(agent.epoch + 1)
    out['agent_steps'] = np.int64(agent.agent_steps)
    if save_ckpt is not None:
        real_save(save_ckpt)    # tensors, numbers and containers only: loads with torch.load(weights_only=True)
    save(tag, **out)


def g6_g8():
    run_agent('g8_train_small_dv3_adamw', NETS['small_dv3'], N=16, T=8, mb=32, mini_epochs=3, epochs=3,
              optim_type='AdamW', dump_first_step=True)
    run_agent('g8_train_small_dv3_adam', NETS['small_dv3'], N=16, T=8, mb=64, mini_epochs=2, epochs=2,
              optim_type='Adam', dump_first_step=True)
    run_agent('g8_train_small_normal_adam', NETS['small_normal'], N=16, T=8, mb=64, mini_epochs=2, epochs=2,
              optim_type='Adam', dump_first_step=True)
    # a mid-size LayerNorm/SiLU net whose widths exercise the MFMA tiling (multiples of 32)
    mid = dict(NETS['full_dv3'], units=[128, 64, 32], obs=64, act=12)
    run_agent('g8_train_mid_dv3_adamw', mid, N=64, T=8, mb=128, mini_epochs=2, epochs=2, optim_type='AdamW',
              dump_first_step=True, params_every_epoch=False)


def g8_cartpole():
    run_agent('g8_train_cartpole_adam', NETS['cartpole'], N=64, T=8, mb=128, mini_epochs=2, epochs=2, optim_type='Adam',
              dump_first_step=True, params_every_epoch=False)


def g8_full():
    """The DFlexAntPPO network itself ([512, 256, 128], LayerNorm + SiLU, sigma head) through the reference's PPO.train()."""
    run_agent('g8_train_full_dv3_adamw', NETS['full_dv3'], N=64, T=8, mb=256, mini_epochs=2, epochs=2, optim_type='AdamW',
              params_every_epoch=False)


def g_lnelu():
    """LayerNorm + ELU towers: the module's outputs and one recorded PPO.train()."""
    g4({'mid_lnelu': LNELU})
    run_agent('g8_train_mid_lnelu_adamw', LNELU, N=64, T=8, mb=128, mini_epochs=2, epochs=2, optim_type='AdamW',
              params_every_epoch=False)


def g16_checkpoint():
    """A checkpoint written by the REFERENCE's PPO.save after one epoch, and the reference's own continuation from it (a fresh
    agent and env, PPO.load, one more epoch).  The product has to load the same file and continue the same way."""
    mid = dict(NETS['full_dv3'], units=[128, 64, 32], obs=64, act=12)
    ckpt = os.path.join(HERE, 'g16_reference_checkpoint.pth')
    run_agent('g16_ckpt_first_epoch', mid, N=64, T=8, mb=128, mini_epochs=2, epochs=1, optim_type='AdamW',
              params_every_epoch=False, save_ckpt=ckpt)
    os.remove(os.path.join(HERE, 'g16_ckpt_first_epoch.npz'))     # only the checkpoint and the continuation are fixtures
    run_agent('g16_ckpt_continued', mid, N=64, T=8, mb=128, mini_epochs=2, epochs=2, optim_type='AdamW',
              params_every_epoch=False, load_ckpt=ckpt)
    print(f'g16_reference_checkpoint.pth: {os.path.getsize(ckpt) / 1024:.1f} KiB')


# ---- G9 / G10: seeds and schedulers -------------------------------------------------------------------
def g9_g10():
    seeds = {}
    for s in (0, 42, 100):
        rng = np.random.RandomState(s)
        seeds[f'seed{s}'] = rng.randint(0, int(1e6), 5)
    sch = AdaptiveScheduler(0.008, 1e-6, 1e-2)
    kls = synth.lcg((40,), 77, 0.04, np.float64) + 0.02
    lr, lrs = 5e-4, []
    for k in kls:
        lr = sch.update(lr, float(k))
        lrs.append(lr)
    lin = LinearScheduler(5e-4, 1e-6, 1000000)
    lin_lrs = [lin.update(s) for s in range(0, 1200000, 100000)]
    cos = [adjust_learning_rate_cos(5e-4, e, 5, s, 4100000) for e in range(5) for s in (0, 1000000, 4000000)]
    save('g9_g10_seeds_sched', kl_lrs=np.array(lrs), lin_lrs=np.array(lin_lrs), cos_lrs=np.array(cos), **seeds)


if __name__ == '__main__':
    only = os.environ.get('GOLDEN_ONLY')                    # add fixtures without rewriting the others
    if only:
        for name in only.split(','):
            {'g4': g4, 'g8_cartpole': g8_cartpole, 'g8_full': g8_full, 'g16_checkpoint': g16_checkpoint, 'lnelu': g_lnelu}[name]()
        sys.exit(0)
    g1_g2()
    g3()
    g4()
    g5()
    g6_g8()
    g8_cartpole()
    g8_full()
    g16_checkpoint()
    g_lnelu()
    g9_g10()
