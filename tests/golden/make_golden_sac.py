"""Golden vectors of the SAC update loop, recorded from the REFERENCE agent (read-only checkout) on CPU:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sac.py [/root/reference]

Uses the stub modules of make_golden.py (omegaconf / wandb / tensorboard are not installed here and never on the numeric
path).  Random draws are injected from synth.py — torch.rand (warm-up actions), Normal.rsample (policy noise) and
torch.randint (replay indices) — so the tests rebuild every input; the fixtures hold outputs only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.argv = sys.argv[:1] + (sys.argv[1:] or ['/root/reference'])

import make_golden as MG  # noqa: E402  (installs the stubs, puts the reference on sys.path; generates nothing on import)
import torch  # noqa: E402

import synth  # noqa: E402
from mineral.agents.sac import sac as ref_sac  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import sac_oracle as S  # noqa: E402  (shape lists only)


def make_cfg(case, logdir):
    N, O, A, au, cu, hp, epochs = synth.SAC_CASES[case]
    mlp = lambda u: dict(units=u, norm_type='LayerNorm', act_type='SiLU')
    sac = dict(multi_gpu=False, num_actors=N, reward_shaper=dict(fn='scale', scale=0.5), max_agent_steps=N * (hp['warm_up'] + epochs * hp['horizon_len']),
               tau=0.05, gamma=0.99, no_tgt_actor=True, handle_timeout=True, update_actor_interval=1, update_targets_interval=1,
               init_alpha=0.7, alpha=None, alpha_optim_kwargs=dict(lr=5e-3), backup_entropy=True, optim_type='AdamW',
               actor_optim_kwargs=dict(lr=5e-4), critic_optim_kwargs=dict(lr=5e-4), max_grad_norm=0.5)
    sac.update(hp)
    cfg = dict(rl_device='cpu', multi_gpu=False, env_render=False, run='train', seed=42,
               task=dict(name='Golden', env=dict(numEnvs=N), env_autoresets=True),
               agent=dict(algo='SAC', print_every=-1, ckpt_every=-1, tracker_len=100, metrics_kwargs={},
                          network=dict(normalize_input=True, encoder=None, actor='Actor',
                                       actor_kwargs=dict(mlp_kwargs=mlp(au), weight_init='orthogonal', tanh_policy=False, fixed_sigma=False,
                                                         dist_kwargs=dict(dist_type='squashed_normal', minlogstd=-5.0, maxlogstd=2.0)),
                                       critic='EnsembleQ', critic_kwargs=dict(n_critics=2, mlp_kwargs=mlp(cu), weight_init='orthogonal')),
                          sac=sac))
    return MG.AttrDict.wrap(cfg)


def run(case):
    N, O, A, au, cu, hp, epochs = synth.SAC_CASES[case]
    cfg = make_cfg(case, f'/tmp/golden_sac_{case}')
    env = synth.SacGoldenEnv(N, O, A, episode_length=5)
    agent = ref_sac.SAC(cfg, logdir=f'/tmp/golden_sac_{case}', accelerator=None, datasets=None, env=env)
    agent.save = lambda *a, **k: None
    agent._checkpoint_save = lambda *a, **k: None
    ocfg = S.SACConfig(O, A, au, cu)
    agent.actor.load_state_dict(synth.make_params(S.actor_shapes(ocfg), 21), strict=True)
    agent.critic.load_state_dict(synth.make_params(S.critic_shapes(ocfg), 22), strict=True)
    agent.critic_target.load_state_dict(agent.critic.state_dict())

    counters = {'rand': 0, 'noise': 0, 'idx': 0}
    orig_rand, orig_randint, orig_rsample = torch.rand, torch.randint, torch.distributions.Normal.rsample

    def fake_rand(*size, **kw):
        shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
        counters['rand'] += 1
        return synth.sac_uniform(shape, counters['rand'] - 1)

    def fake_randint(high, size, **kw):
        counters['idx'] += 1
        return synth.sac_indices(int(high), int(size[0]), counters['idx'] - 1)

    def fake_rsample(self, sample_shape=torch.Size()):
        counters['noise'] += 1
        return self.loc + self.scale * synth.sac_noise(self.loc.shape, counters['noise'] - 1)

    out = {}
    orig_update = agent.update_net   # timer-wrapped

    def spy_update(memory):
        res = orig_update(memory)
        e = agent.epoch
        for k, v in res.items():
            out[f'e{e}_{k.replace("/", "_")}'] = torch.stack([torch.as_tensor(x).detach().float() for x in v]).numpy()
        for k, v in agent.actor.state_dict().items():
            out[f'e{e}_actor.{k}'] = v.detach().numpy().copy()
        for k, v in agent.critic.state_dict().items():
            out[f'e{e}_critic.{k}'] = v.detach().numpy().copy()
        if e == epochs - 1:
            for k, v in agent.critic_target.state_dict().items():
                out[f'e{e}_critic_target.{k}'] = v.detach().numpy().copy()
        out[f'e{e}_log_alpha'] = agent.log_alpha.detach().numpy().copy()
        out[f'e{e}_counters'] = np.array([counters['rand'], counters['noise'], counters['idx']], dtype=np.int64)
        for k, v in agent.obs_rms['obs'].state_dict().items():
            out[f'e{e}_obs_rms.{k}'] = v.numpy().copy()
        out[f'e{e}_memory_state'] = np.array([memory.next_p, memory.cur_capacity, int(memory.if_full)], dtype=np.int64)
        k = memory.cur_capacity
        out[f'e{e}_memory_reward'] = memory.buf_reward[:k].numpy().copy()
        out[f'e{e}_memory_done'] = memory.buf_done[:k].numpy().astype(np.uint8).copy()
        if e == 0:
            out['e0_memory_obs'] = memory.buf_obs['obs'][:k].numpy().copy()
            out['e0_memory_action'] = memory.buf_action[:k].numpy().copy()
            out['e0_memory_next_obs'] = memory.buf_next_obs['obs'][:k].numpy().copy()
        return res

    agent.update_net = spy_update
    torch.rand, torch.randint, torch.distributions.Normal.rsample = fake_rand, fake_randint, fake_rsample
    try:
        agent.train()
    finally:
        torch.rand, torch.randint, torch.distributions.Normal.rsample = orig_rand, orig_randint, orig_rsample
    assert agent.epoch == epochs - 1, (agent.epoch, epochs)
    path = os.path.join(HERE, f'g12_sac_{case}.npz')
    np.savez_compressed(path, **out)
    print(case, 'wrote', os.path.getsize(path), 'bytes,', len(out), 'arrays; draws', counters)


if __name__ == '__main__':
    for case in synth.SAC_CASES:
        run(case)
