"""Helpers shared by the replay buffers: the five tensors of a transition set and their C view."""

import torch

from .. import _hip


def single_key(obs_space):
    """The HIP replay path handles one flat observation key (like the PPO path); anything else is refused."""
    if len(obs_space) != 1:
        raise NotImplementedError(f'replay buffers on the HIP path take one observation key, got {sorted(obs_space)}')
    (key, shape), = obs_space.items()
    shape = tuple(shape)
    if len(shape) != 1 or shape[0] % 4:
        raise NotImplementedError(f"observation '{key}' must be flat with a length that is a multiple of 4, got {shape}")
    return key, int(shape[0])


def alloc(rows, obs_dim, act_dim, device, done_dtype=torch.float32):
    """obs, action, reward, next_obs, done of `rows` transitions (simple_replay.py:4-16 with flat leading dimension)."""
    return (torch.empty(rows, obs_dim, dtype=torch.float32, device=device), torch.empty(rows, act_dim, dtype=torch.float32, device=device),
            torch.empty(rows, 1, dtype=torch.float32, device=device), torch.empty(rows, obs_dim, dtype=torch.float32, device=device),
            torch.empty(rows, 1, dtype=done_dtype, device=device))


def view(tensors, obs_dim, act_dim):
    """C struct over five contiguous device tensors.  The caller keeps the tensors alive."""
    obs, action, reward, next_obs, done = tensors
    for t in tensors:
        if not t.is_contiguous():
            raise ValueError('replay tensors must be contiguous')
    return _hip.Transitions(obs.data_ptr(), action.data_ptr(), reward.data_ptr(), next_obs.data_ptr(), done.data_ptr(), obs_dim, act_dim)


def require_cuda(device):
    if torch.device(device).type != 'cuda':
        raise _hip.HipLibraryError(f"the replay buffers run on the HIP device only (device='{device}'); there is no CPU fallback")


def stream_of(device):
    return torch.cuda.current_stream(torch.device(device)).cuda_stream
