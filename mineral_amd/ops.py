"""The entry points of libmineral_ppo.so as PyTorch custom operators: `torch.ops.mineral.*`.

Each operator is a thin dispatcher-registered wrapper (torch.library, CUDA key = the ROCm device) over one function of the
C ABI in include/mineral_ppo.h: it takes and returns tensors, launches on the current stream of the tensors' device and
raises on a non-zero status.  Registration happens on import of this module; nothing is registered for the CPU key, so
calling an operator with CPU tensors fails in the dispatcher instead of falling back.

    import mineral_amd.ops
    torch.ops.mineral.gae_finish(rewards_tn, values_tn, dones_tn, neglogp_tn, last_values, 0.99, 0.95, ...)

Network description for the operators that run the towers: `net_i` = [obs_dim, act_dim, dist_type, tower_kind, fixed_sigma,
units...] (dist_type 0 normal / 1 dreamerv3_normal; tower_kind 0 LayerNorm+SiLU / 1 ELU / 2 LayerNorm+ELU) and `net_f` = [minstd, maxstd, ln_eps].
"""

import torch

from . import _hip

_lib = torch.library.Library('mineral', 'DEF')


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _net(net_i, net_f):
    n = _hip.Net()
    n.obs_dim, n.act_dim, n.dist_type, n.tower_kind, n.fixed_sigma = (int(x) for x in net_i[:5])
    units = [int(u) for u in net_i[5:]]
    if not 1 <= len(units) <= _hip.MAX_LAYERS:
        raise ValueError(f'net_i carries {len(units)} layer widths; 1..{_hip.MAX_LAYERS} are supported')
    n.n_layers = len(units)
    for i, u in enumerate(units):
        n.units[i] = u
    n.minstd, n.maxstd, n.ln_eps = (float(x) for x in net_f[:3])
    return n


def _rms(mean, var, count, eps, lo, hi):
    return None if mean is None else _hip.Rms(mean.data_ptr(), var.data_ptr(), count.data_ptr(), mean.numel(), eps, lo, hi)


def _rollout(obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat):
    """mrl_rollout over the caller's tensors: obs [N][T][O], actions / mu / sigma [N][T][A], the four [T][N] staging tensors and
    `flat` [4][N*T] = values, returns, advantages, neglogp."""
    T, N = neglogp_tn.shape
    return _hip.Rollout(N, T, obs.data_ptr(), actions.data_ptr(), mu.data_ptr(), sigma.data_ptr(), neglogp_tn.data_ptr(), values_tn.data_ptr(),
                        rewards_tn.data_ptr(), dones_tn.data_ptr(), flat[0].data_ptr(), flat[1].data_ptr(), flat[2].data_ptr(), flat[3].data_ptr())


_ROLL = 'Tensor(r0!) obs, Tensor(r1!) actions, Tensor(r2!) mu, Tensor(r3!) sigma, Tensor(r4!) neglogp_tn, Tensor(r5!) values_tn, Tensor(r6!) rewards_tn, Tensor(r7!) dones_tn, Tensor(r8!) flat'

_lib.define('workspace_bytes(int[] net_i, float[] net_f, int rows) -> int')
_lib.define('param_count(int[] net_i, float[] net_f) -> int')
_lib.define(f'rollout_write({_ROLL}, int t, Tensor rewards, Tensor dones, Tensor? time_outs, float reward_scale, float gamma) -> ()')
_lib.define(f'gae_finish({_ROLL}, Tensor last_values, float gamma, float tau, Tensor(a!)? rms_mean, Tensor(b!)? rms_var, Tensor(c!)? rms_count, '
            'float rms_eps, float clamp_lo, float clamp_hi, Tensor(d!) stats, Tensor(e!) workspace) -> ()')
_lib.define('welford_update_(Tensor(a!) mean, Tensor(b!) var, Tensor(c!) count, float eps, Tensor batch_mean, Tensor batch_var, float batch_count, '
            'Tensor(d!)? mean_f32, Tensor(e!)? std_f32) -> ()')
_lib.define('normalize_clamp(Tensor x, Tensor mean, Tensor var, Tensor count, float eps, float clamp_lo, float clamp_hi, bool unnormalize) -> Tensor')
_lib.define(f'policy_act(int[] net_i, float[] net_f, Tensor params, Tensor obs_in, Tensor? noise, int t, {_ROLL}, Tensor(a!)? actions_clamped, '
            'Tensor(b!)? last_values, Tensor? obs_mean, Tensor? obs_var, Tensor? obs_count, float obs_eps, Tensor? val_mean, Tensor? val_var, '
            'Tensor? val_count, float val_eps, float clamp_lo, float clamp_hi, Tensor(c!) workspace) -> ()')
_lib.define(f'ppo_grad(int[] net_i, float[] net_f, Tensor params, float[] hp_f, int[] hp_i, {_ROLL}, int row0, int mb, Tensor? obs_mean_f32, '
            'Tensor? obs_std_f32, float clamp_lo, float clamp_hi, Tensor(a!) grads, Tensor(b!) metrics, Tensor(c!) workspace) -> ()')
_lib.define('clip_adam_step_(Tensor(a!) params, Tensor grads, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, float beta1, float beta2, float eps, '
            'float weight_decay, bool adamw, int step, float max_norm, float grad_scale, Tensor(d!)? gnorm, Tensor(e!) scratch) -> ()')


def _workspace_bytes(net_i, net_f, rows):
    r = int(_hip.lib().mrl_workspace_bytes(_net(net_i, net_f), rows))
    if r < 0:
        _hip.check(r, 'mrl_workspace_bytes')
    return r


def _param_count(net_i, net_f):
    r = int(_hip.lib().mrl_param_count(_net(net_i, net_f)))
    if r < 0:
        _hip.check(r, 'mrl_param_count')
    return r


def _rollout_write(obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat, t, rewards, dones, time_outs, reward_scale, gamma):
    c = _rollout(obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat)
    d8 = dones.view(torch.uint8) if dones.dtype == torch.bool else dones.to(torch.uint8)
    to8 = None if time_outs is None else (time_outs.view(torch.uint8) if time_outs.dtype == torch.bool else time_outs.to(torch.uint8))
    _hip.check(_hip.lib().mrl_rollout_write(c, t, _hip.ptr(rewards.contiguous()), _hip.ptr(d8.contiguous()), _hip.ptr(to8), reward_scale, gamma,
                                            _stream(rewards)), 'mrl_rollout_write')


def _gae_finish(obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat, last_values, gamma, tau, rms_mean, rms_var, rms_count,
                rms_eps, clamp_lo, clamp_hi, stats, workspace):
    c = _rollout(obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat)
    _hip.check(_hip.lib().mrl_gae_finish(c, _hip.ptr(last_values.contiguous()), gamma, tau, _rms(rms_mean, rms_var, rms_count, rms_eps, clamp_lo, clamp_hi),
                                         _hip.ptr(stats), _hip.ptr(workspace), workspace.numel(), _stream(stats)), 'mrl_gae_finish')


def _welford_update(mean, var, count, eps, batch_mean, batch_var, batch_count, mean_f32, std_f32):
    _hip.check(_hip.lib().mrl_rms_merge(_rms(mean, var, count, eps, -5.0, 5.0), _hip.ptr(batch_mean), _hip.ptr(batch_var), batch_count, _hip.ptr(mean_f32),
                                        _hip.ptr(std_f32), _stream(mean)), 'mrl_rms_merge')


def _normalize_clamp(x, mean, var, count, eps, clamp_lo, clamp_hi, unnormalize):
    x = x.contiguous()
    y = torch.empty_like(x)
    _hip.check(_hip.lib().mrl_rms_normalize(_rms(mean, var, count, eps, clamp_lo, clamp_hi), _hip.ptr(x), _hip.ptr(y), x.numel() // mean.numel(),
                                            int(unnormalize), _stream(x)), 'mrl_rms_normalize')
    return y


def _policy_act(net_i, net_f, params, obs_in, noise, t, obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat, actions_clamped,
                last_values, obs_mean, obs_var, obs_count, obs_eps, val_mean, val_var, val_count, val_eps, clamp_lo, clamp_hi, workspace):
    c = _rollout(obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat)
    _hip.check(_hip.lib().mrl_policy_act(_net(net_i, net_f), _hip.ptr(params), _rms(obs_mean, obs_var, obs_count, obs_eps, clamp_lo, clamp_hi),
                                         _rms(val_mean, val_var, val_count, val_eps, clamp_lo, clamp_hi), _hip.ptr(obs_in.contiguous()), _hip.ptr(noise), t, c,
                                         _hip.ptr(actions_clamped), _hip.ptr(last_values), _hip.ptr(workspace), workspace.numel(), _stream(params)),
               'mrl_policy_act')


def _ppo_grad(net_i, net_f, params, hp_f, hp_i, obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat, row0, mb, obs_mean_f32,
              obs_std_f32, clamp_lo, clamp_hi, grads, metrics, workspace):
    c = _rollout(obs, actions, mu, sigma, neglogp_tn, values_tn, rewards_tn, dones_tn, flat)
    hp = _hip.PpoHp(*(float(x) for x in hp_f[:4]), *(int(x) for x in hp_i[:3]))     # e_clip, critic_coef, entropy_coef, bounds_loss_coef | bounds_type, clip_value_loss, use_smooth_clamp
    _hip.check(_hip.lib().mrl_ppo_grad(_net(net_i, net_f), _hip.ptr(params), hp, c, row0, mb, _hip.ptr(obs_mean_f32), _hip.ptr(obs_std_f32), clamp_lo, clamp_hi,
                                       _hip.ptr(grads), _hip.ptr(metrics), _hip.ptr(workspace), workspace.numel(), _stream(params)), 'mrl_ppo_grad')


def _clip_adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, adamw, step, max_norm, grad_scale, gnorm, scratch):
    _hip.check(_hip.lib().mrl_adam_step(_hip.ptr(params), _hip.ptr(grads), _hip.ptr(exp_avg), _hip.ptr(exp_avg_sq), params.numel(), lr, beta1, beta2, eps,
                                        weight_decay, int(adamw), step, max_norm, grad_scale, _hip.ptr(gnorm), _hip.ptr(scratch), _stream(params)),
               'mrl_adam_step')


_lib.impl('workspace_bytes', _workspace_bytes, 'CompositeExplicitAutograd')     # host-only queries: no tensor arguments
_lib.impl('param_count', _param_count, 'CompositeExplicitAutograd')
for _name, _fn in (('rollout_write', _rollout_write), ('gae_finish', _gae_finish), ('welford_update_', _welford_update), ('normalize_clamp', _normalize_clamp),
                   ('policy_act', _policy_act), ('ppo_grad', _ppo_grad), ('clip_adam_step_', _clip_adam_step)):
    _lib.impl(_name, _fn, 'CUDA')

OPS = ('workspace_bytes', 'param_count', 'rollout_write', 'gae_finish', 'welford_update_', 'normalize_clamp', 'policy_act', 'ppo_grad', 'clip_adam_step_')
