"""ctypes binding of `libmineral_ppo.so` (the C ABI declared in include/mineral_ppo.h and include/mineral_replay.h).

The library is the product: there is no Python/torch fallback.  If it has not been built (`make` at the repo
root) importing a function from here raises immediately.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MRL_LIB selects another build of the same library (diagnostic builds under tools/bin); the default is the in-tree product
LIB_PATH = os.environ.get('MRL_LIB') or os.path.join(_HERE, 'libmineral_ppo.so')

MAX_LAYERS = 4
NMETRIC = 12
DIST_TYPES = {'normal': 0, 'dreamerv3_normal': 1}
METRIC_NAMES = ('loss/actor', 'loss/entropy', 'loss/bounds', 'kl', 'clip_frac', 'loss/critic', 'explained_var', 'loss/total')


class HipLibraryError(RuntimeError):
    pass


class Net(C.Structure):
    _fields_ = [('obs_dim', C.c_int32), ('act_dim', C.c_int32), ('n_layers', C.c_int32), ('units', C.c_int32 * MAX_LAYERS),
                ('dist_type', C.c_int32), ('minstd', C.c_float), ('maxstd', C.c_float), ('ln_eps', C.c_float),
                ('tower_kind', C.c_int32), ('fixed_sigma', C.c_int32)]


class Rms(C.Structure):
    _fields_ = [('mean', C.c_void_p), ('var', C.c_void_p), ('count', C.c_void_p), ('size', C.c_int32), ('eps', C.c_float),
                ('clamp_lo', C.c_float), ('clamp_hi', C.c_float)]


class Rollout(C.Structure):
    _fields_ = [('num_envs', C.c_int64), ('horizon', C.c_int32), ('obs', C.c_void_p), ('actions', C.c_void_p),
                ('mu', C.c_void_p), ('sigma', C.c_void_p), ('neglogp_tn', C.c_void_p), ('values_tn', C.c_void_p),
                ('rewards_tn', C.c_void_p), ('dones_tn', C.c_void_p), ('values', C.c_void_p), ('returns', C.c_void_p),
                ('advantages', C.c_void_p), ('neglogp', C.c_void_p)]


class PpoHp(C.Structure):
    _fields_ = [('e_clip', C.c_float), ('critic_coef', C.c_float), ('entropy_coef', C.c_float), ('bounds_loss_coef', C.c_float),
                ('bounds_type', C.c_int32), ('clip_value_loss', C.c_int32), ('use_smooth_clamp', C.c_int32)]


class Transitions(C.Structure):
    """mrl_transitions: structure-of-arrays view of a set of transitions (include/mineral_replay.h)."""
    _fields_ = [('obs', C.c_void_p), ('action', C.c_void_p), ('reward', C.c_void_p), ('next_obs', C.c_void_p), ('done', C.c_void_p),
                ('obs_dim', C.c_int32), ('act_dim', C.c_int32)]


NSTEP_MAX = 7
SAC_NMETRIC = 8


class SacNet(C.Structure):
    """mrl_sac_net (include/mineral_sac.h)."""
    _fields_ = [('obs_dim', C.c_int32), ('act_dim', C.c_int32), ('actor_layers', C.c_int32), ('actor_units', C.c_int32 * MAX_LAYERS),
                ('critic_layers', C.c_int32), ('critic_units', C.c_int32 * MAX_LAYERS), ('minlogstd', C.c_float), ('maxlogstd', C.c_float),
                ('ln_eps', C.c_float), ('actor_kind', C.c_int32), ('critic_kind', C.c_int32)]


def make_sac_net(obs_dim, act_dim, actor_units, critic_units, minlogstd=-5.0, maxlogstd=2.0, ln_eps=1e-5, actor_kind=0, critic_kind=0):
    """actor_kind / critic_kind: MRL_TOWER_* (TOWER_KINDS[(norm_type, act_type)]) of the two mlp_kwargs."""
    if len(actor_units) > MAX_LAYERS or len(critic_units) > MAX_LAYERS:
        raise NotImplementedError(f'at most {MAX_LAYERS} layers per tower')
    au = (C.c_int32 * MAX_LAYERS)(*actor_units)
    cu = (C.c_int32 * MAX_LAYERS)(*critic_units)
    return SacNet(obs_dim, act_dim, len(actor_units), au, len(critic_units), cu, minlogstd, maxlogstd, ln_eps, int(actor_kind), int(critic_kind))


class CriticNet(C.Structure):
    """mrl_critic_net (include/mineral_shac.h)."""
    _fields_ = [('obs_dim', C.c_int32), ('n_critics', C.c_int32), ('n_layers', C.c_int32), ('units', C.c_int32 * MAX_LAYERS), ('ln_eps', C.c_float),
                ('act_type', C.c_int32)]


CRITIC_ACTS = {'SiLU': 0, 'ELU': 1}


def make_critic_net(obs_dim, n_critics, units, ln_eps=1e-5, act_type='SiLU'):
    if len(units) > MAX_LAYERS:
        raise NotImplementedError(f'at most {MAX_LAYERS} layers per tower')
    if act_type not in CRITIC_ACTS:
        raise NotImplementedError(f'act_type={act_type}')
    return CriticNet(obs_dim, n_critics, len(units), (C.c_int32 * MAX_LAYERS)(*units), ln_eps, CRITIC_ACTS[act_type])


_P = C.c_void_p
# int reduce(void* ctx, int64_t step_index, void* stream): the gradient all-reduce hook of mrl_ppo_mini_epoch_reduce
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)
_SIGNATURES = {
    'mrl_last_error': (C.c_char_p, []),
    'mrl_version': (C.c_int, []),
    'mrl_param_count': (C.c_int64, [C.POINTER(Net)]),
    'mrl_param_offsets': (C.c_int, [C.POINTER(Net), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'mrl_workspace_bytes': (C.c_int64, [C.POINTER(Net), C.c_int64]),
    'mrl_policy_act': (C.c_int, [C.POINTER(Net), _P, C.POINTER(Rms), C.POINTER(Rms), _P, _P, C.c_int32, C.POINTER(Rollout), _P, _P,
                                 _P, C.c_int64, _P]),
    'mrl_policy_mean': (C.c_int, [C.POINTER(Net), _P, C.POINTER(Rms), _P, C.c_int64, _P, _P, C.c_int64, _P]),
    'mrl_info_track': (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P]),
    'mrl_rollout_write': (C.c_int, [C.POINTER(Rollout), C.c_int32, _P, _P, _P, C.c_double, C.c_double, _P]),
    'mrl_episode_update': (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, C.c_int32, _P, _P, _P]),
    'mrl_gae_finish': (C.c_int, [C.POINTER(Rollout), _P, C.c_double, C.c_double, C.POINTER(Rms), _P, _P, C.c_int64, _P]),
    'mrl_gae_finish_static': (C.c_int, [C.POINTER(Rollout), _P, C.c_double, C.c_double, C.POINTER(Rms), _P, _P]),
    'mrl_gae_scan': (C.c_int, [C.POINTER(Rollout), _P, C.c_double, C.c_double, _P, _P, C.c_int64, _P]),
    'mrl_gae_normalize': (C.c_int, [C.POINTER(Rollout), _P, C.c_double, C.POINTER(Rms), _P, _P]),
    'mrl_batch_moments': (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    'mrl_rms_merge': (C.c_int, [C.POINTER(Rms), _P, _P, C.c_double, _P, _P, _P]),
    'mrl_rms_normalize': (C.c_int, [C.POINTER(Rms), _P, _P, C.c_int64, C.c_int32, _P]),
    'mrl_ppo_grad': (C.c_int, [C.POINTER(Net), _P, C.POINTER(PpoHp), C.POINTER(Rollout), C.c_int64, C.c_int64, _P, _P, C.c_float,
                               C.c_float, _P, _P, _P, C.c_int64, _P]),
    'mrl_adam_step': (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32,
                                C.c_int64, C.c_float, C.c_float, _P, _P, _P]),
    'mrl_ppo_step': (C.c_int, [C.POINTER(Net), _P, C.POINTER(PpoHp), C.POINTER(Rollout), C.c_int64, C.c_int64, C.POINTER(Rms), _P, _P, C.c_double,
                               _P, _P, C.c_float, C.c_float, _P, _P, _P, C.c_int64, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double,
                               C.c_double, C.c_int32, C.c_int64, C.c_float, _P, _P, _P]),
    'mrl_ppo_mini_epoch': (C.c_int, [C.POINTER(Net), _P, C.POINTER(PpoHp), C.POINTER(Rollout), C.c_int64, C.c_int64, C.POINTER(Rms), _P, _P,
                                     C.c_double, _P, _P, C.c_float, C.c_float, _P, _P, _P, C.c_int64, _P, _P, C.c_double, C.c_double, C.c_double,
                                     C.c_double, C.c_double, C.c_int32, C.c_int64, C.c_float, _P, _P, _P]),
    'mrl_ppo_mini_epoch_reduce': (C.c_int, [C.POINTER(Net), _P, C.POINTER(PpoHp), C.POINTER(Rollout), C.c_int64, C.c_int64, C.POINTER(Rms), _P, _P,
                                            C.c_double, _P, _P, C.c_float, C.c_float, _P, _P, _P, C.c_int64, _P, _P, C.c_double, C.c_double,
                                            C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int64, C.c_float, _P, _P, _P, C.c_float, REDUCE_FN,
                                            _P]),
    'mrl_nstep_emit': (C.c_int, [C.POINTER(Transitions), C.c_int64, C.c_int32, C.POINTER(Transitions), C.c_int64, C.c_int32, _P,
                                 C.POINTER(Transitions), _P]),
    'mrl_replay_write': (C.c_int, [C.POINTER(Transitions), C.c_int64, C.c_int64, C.POINTER(Transitions), C.c_int64, _P]),
    'mrl_replay_gather': (C.c_int, [C.POINTER(Transitions), C.c_int64, _P, C.c_int64, C.POINTER(Transitions), _P]),
    'mrl_sac_param_count': (C.c_int64, [C.POINTER(SacNet), C.c_int32]),
    'mrl_sac_param_offsets': (C.c_int, [C.POINTER(SacNet), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'mrl_sac_workspace_bytes': (C.c_int64, [C.POINTER(SacNet), C.c_int64]),
    'mrl_sac_actor_sample': (C.c_int, [C.POINTER(SacNet), _P, _P, C.c_int64, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, C.c_int64, _P]),
    'mrl_sac_critic_grad': (C.c_int, [C.POINTER(SacNet), _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.POINTER(Transitions), C.c_int64, _P,
                                      C.c_float, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    'mrl_sac_actor_grad': (C.c_int, [C.POINTER(SacNet), _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, C.c_int64, _P, C.c_float, _P, _P, _P, _P,
                                     C.c_int64, _P]),
    'mrl_polyak': (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    'mrl_critic_param_count': (C.c_int64, [C.POINTER(CriticNet)]),
    'mrl_critic_param_offsets': (C.c_int, [C.POINTER(CriticNet), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'mrl_critic_workspace_bytes': (C.c_int64, [C.POINTER(CriticNet), C.c_int64]),
    'mrl_tdlambda_targets': (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_int32, _P, _P]),
    'mrl_tdlambda_targets_entropy': (C.c_int, [_P, _P, _P, _P, C.c_double, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_int32, _P, _P]),
    'mrl_critic_values': (C.c_int, [C.POINTER(CriticNet), _P, _P, C.c_int64, _P, _P, C.c_float, C.c_float, _P, _P, C.c_int64, _P]),
    'mrl_critic_mse_grad': (C.c_int, [C.POINTER(CriticNet), _P, _P, _P, C.c_int64, _P, _P, C.c_float, C.c_float, _P, _P, _P, C.c_int64, _P]),
    'mrl_lerp_params': (C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_float, _P]),
    'mrl_profile_enable': (C.c_int, [C.c_int]),
    'mrl_profile_read_all': (C.c_int, [C.c_char_p, C.c_int32]),
    'mrl_profile_read': (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_char_p, C.c_int32]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load the shared library once; raise HipLibraryError (never fall back) if it is missing or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(f'{LIB_PATH} not found: build it with `make` at the repository root '
                                  '(hipcc --offload-arch=gfx950); there is no CPU fallback')
        # torch ships its own HIP runtime; load it first so that the library binds to the same runtime instance
        # that owns the tensors' device memory and streams
        import torch  # noqa: F401
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise HipLibraryError(f'cannot load {LIB_PATH}: {e}') from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise HipLibraryError(f'{LIB_PATH} does not export {name}') from e
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().mrl_last_error().decode('utf-8', 'replace')
        if status == -1:
            raise ValueError(f'{what}: {msg}')
        raise HipLibraryError(f'{what} failed ({status}): {msg}')


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def require_current_device(device):
    """The library launches its kernels on the CURRENT HIP device and never selects one itself.  Pointers that live on
    another GPU would be dereferenced by a kernel running on the current one (a memory fault without peer access), so the
    host classes call this once at construction and refuse a device that is not current."""
    import torch
    device = torch.device(device)
    if device.type != 'cuda':
        raise HipLibraryError(f'device {device} is not a GPU; the HIP path has no CPU fallback')
    index = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_device()
    if index != cur:
        raise HipLibraryError(f'tensors live on cuda:{index} but the current device is cuda:{cur}: call torch.cuda.set_device({index}) '
                              'before constructing the agent (the launcher does)')
    return index


# (norm_type, act_type) of mineral/nets/mlp.py -> MRL_TOWER_*
TOWER_KINDS = {('LayerNorm', 'SiLU'): 0, (None, 'ELU'): 1, ('LayerNorm', 'ELU'): 2, (None, 'SiLU'): 3, ('LayerNorm', 'ReLU'): 4, (None, 'ReLU'): 5,
               ('LayerNorm', 'Tanh'): 6, (None, 'Tanh'): 7}


def make_net(obs_dim, act_dim, units, dist_type, minstd=1.0, maxstd=1.0, ln_eps=1e-5, tower_kind=0, fixed_sigma=False):
    if len(units) > MAX_LAYERS:
        raise ValueError(f'at most {MAX_LAYERS} hidden layers are supported, got {len(units)}')
    if dist_type not in DIST_TYPES:
        raise NotImplementedError(dist_type)
    n = Net()
    n.obs_dim, n.act_dim, n.n_layers = obs_dim, act_dim, len(units)
    for i, u in enumerate(units):
        n.units[i] = int(u)
    n.dist_type, n.minstd, n.maxstd, n.ln_eps = DIST_TYPES[dist_type], minstd, maxstd, ln_eps
    n.tower_kind, n.fixed_sigma = int(tower_kind), int(bool(fixed_sigma))
    return n
