"""Golden vectors of the SHAC critic side, recorded from the REFERENCE (read-only checkout) on CPU:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_shac.py [/root/reference]

The reference's own `SHAC.compute_target_values`, `CriticDataset` and `SHAC.update_critic` (on `EnsembleCritic` and on the single
`Critic`) are called on a namespace object
that carries the attributes those methods read (building the whole agent needs a differentiable simulator).  Inputs come
from synth.py; the fixture holds outputs only.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.argv = sys.argv[:1] + (sys.argv[1:] or ['/root/reference'])

import make_golden as MG  # noqa: E402,F401  (stubs + reference on sys.path)
import torch  # noqa: E402

import synth  # noqa: E402
from mineral.agents.diffrl import models as ref_models  # noqa: E402
from mineral.agents.diffrl import shac as ref_shac  # noqa: E402
from mineral.agents.diffrl.utils import CriticDataset, soft_update  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import shac_oracle as S  # noqa: E402  (shape lists only)

out = {}
for case, (N, T, O, units, n_critics, n_batches, iters, optim_type, lr, max_norm) in synth.SHAC_CASES.items():
    obs, rew, done, nv = synth.shac_inputs(case)
    for method in ('td-lambda', 'one-step'):
        ns = types.SimpleNamespace(critic_method=method, rew_buf=rew, gamma=0.99, lam=0.95, next_values=nv, done_mask=done, num_envs=N, device='cpu',
                                   horizon_len=T, target_values=torch.zeros(T, N))
        ref_shac.SHAC.compute_target_values(ns)
        out[f'{case}_targets_{method}'] = ns.target_values.numpy().copy()
    targets = torch.from_numpy(out[f'{case}_targets_td-lambda'])
    critic = ref_models.EnsembleCritic(O, 4, n_critics=n_critics, mlp_kwargs=dict(units=units, norm_type='LayerNorm', act_type='SiLU'),
                                       weight_init='orthogonal')
    critic.load_state_dict(synth.make_params(S.critic_shapes(O, units, n_critics), 61), strict=True)
    OptimCls = getattr(torch.optim, optim_type)
    optim = OptimCls(critic.parameters(), lr=lr, betas=(0.7, 0.95))
    cfgd = MG.AttrDict.wrap(dict(truncate_grads=True, max_grad_norm=max_norm))
    ns = types.SimpleNamespace(critic_iterations=iters, critic_optim=optim, critic=critic, encoder=lambda o: o['obs'], shac_config=cfgd)
    ns.compute_critic_loss = lambda o, t, **kw: ref_shac.SHAC.compute_critic_loss(ns, o, t, **kw)
    batch_size = N * T // n_batches
    dataset = CriticDataset(batch_size, {'obs': obs}, targets, drop_last=False)
    out[f'{case}_dataset_len'] = np.array([len(dataset), batch_size], dtype=np.int64)
    o_last, t_last = dataset[len(dataset) - 1]
    out[f'{case}_last_batch_targets'] = t_last.numpy().copy()
    res = ref_shac.SHAC.update_critic(ns, dataset)
    for k, v in res.items():
        out[f'{case}_{k.replace("/", "_")}'] = torch.stack(v).detach().numpy()
    for k, v in critic.state_dict().items():
        out[f'{case}_critic.{k}'] = v.numpy().copy()
    target = ref_models.EnsembleCritic(O, 4, n_critics=n_critics, mlp_kwargs=dict(units=units, norm_type='LayerNorm', act_type='SiLU'))
    target.load_state_dict(synth.make_params(S.critic_shapes(O, units, n_critics), 62), strict=True)
    with torch.no_grad():
        soft_update(critic, target, 0.2)
    for k, v in target.state_dict().items():
        out[f'{case}_target.{k}'] = v.numpy().copy()

# entropy-augmented targets (shac.py:893-916), every combination of the two entropy transforms, both methods
ent_out = {}
for case, (N, T, *_rest) in synth.SHAC_CASES.items():
    obs, rew, done, nv = synth.shac_inputs(case)
    logp = synth.gaussianish((T, N), 23) * 1.5 - 2.0
    for method in ('td-lambda', 'one-step'):
        for offset in (False, True):
            for scale in (False, True):
                ns = types.SimpleNamespace(critic_method=method, rew_buf=rew, gamma=0.99, lam=0.95, next_values=nv, done_mask=done, num_envs=N,
                                           device='cpu', horizon_len=T, target_values=torch.zeros(T, N), use_distr_ent=False, logprobs=logp,
                                           offset_by_target_entropy=offset, scale_by_target_entropy=scale, target_entropy=-3.0,
                                           with_autoent=False, entropy_coef=0.05)
                ref_shac.SHAC.compute_target_values_with_entropy(ns)
                ent_out[f'{case}_{method}_o{int(offset)}s{int(scale)}'] = ns.target_values.numpy().copy()
ent_path = os.path.join(HERE, 'g14_shac_entropy_targets.npz')
np.savez_compressed(ent_path, **ent_out)
print('wrote', ent_path, os.path.getsize(ent_path), 'bytes,', len(ent_out), 'arrays')

# the single LayerNorm + ELU `Critic` of SHAC v1 (models.py:149-178) through the same update_critic
v1_out = {}
for case, (N, T, O, units, _one, n_batches, iters, optim_type, lr, max_norm) in synth.SHAC_V1_CASES.items():
    obs, rew, done, nv = synth.shac_inputs(case)
    ns = types.SimpleNamespace(critic_method='td-lambda', rew_buf=rew, gamma=0.99, lam=0.95, next_values=nv, done_mask=done, num_envs=N, device='cpu',
                               horizon_len=T, target_values=torch.zeros(T, N))
    ref_shac.SHAC.compute_target_values(ns)
    targets = ns.target_values.clone()
    critic = ref_models.Critic(O, 4, mlp_kwargs=dict(units=units, norm_type='LayerNorm', act_type='ELU'), weight_init='orthogonal')
    critic.load_state_dict(synth.make_params(S.critic_shapes(O, units, 1, single=True), 63), strict=True)
    with torch.no_grad():
        v1_out[f'{case}_values0'] = critic(obs.reshape(-1, O)).numpy().copy()
    optim = getattr(torch.optim, optim_type)(critic.parameters(), lr=lr, betas=(0.7, 0.95))
    cfgd = MG.AttrDict.wrap(dict(truncate_grads=True, max_grad_norm=max_norm))
    ns = types.SimpleNamespace(critic_iterations=iters, critic_optim=optim, critic=critic, encoder=lambda o: o['obs'], shac_config=cfgd)
    ns.compute_critic_loss = lambda o, t, **kw: ref_shac.SHAC.compute_critic_loss(ns, o, t, **kw)
    dataset = CriticDataset(N * T // n_batches, {'obs': obs}, targets, drop_last=False)
    res = ref_shac.SHAC.update_critic(ns, dataset)
    for k, v in res.items():
        v1_out[f'{case}_{k.replace("/", "_")}'] = torch.stack(v).detach().numpy()
    for k, v in critic.state_dict().items():
        v1_out[f'{case}_critic.{k}'] = v.numpy().copy()
v1_path = os.path.join(HERE, 'g15_shac_v1_critic.npz')
np.savez_compressed(v1_path, **v1_out)
print('wrote', v1_path, os.path.getsize(v1_path), 'bytes,', len(v1_out), 'arrays')
if os.environ.get('GOLDEN_ONLY') in ('g14', 'g15'):
    sys.exit(0)

path = os.path.join(HERE, 'g13_shac_critic.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')
