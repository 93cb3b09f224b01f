"""The entry point that allocates nothing (`mrl_gae_finish_static`, scratch in a device array built into the library)
against `mrl_gae_finish` with a caller workspace: bit-identical outputs on two calls in a row, the second on a smaller
rollout (stale scratch contents of the first call must not be consumed)."""

import pytest
import torch

import synth

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _run(N, T, static, seed):
    from mineral_amd import _hip
    from mineral_amd.agents.ppo.experience import ExperienceBuffer
    from mineral_amd.common.normalizers import RunningMeanStd
    L = _hip.lib()
    buf = ExperienceBuffer(N, T, N * T, N * T, {'obs': (8,)}, 2, DEV)
    s = buf.storage_dict
    s['rewards'].copy_(synth.gaussianish((T, N, 1), seed))
    s['values'].copy_(synth.gaussianish((T, N, 1), seed + 1))
    s['dones'].copy_((synth.lcg_t((T, N), seed + 2) > 0.8).to(torch.uint8))
    s['neglogp'].copy_(synth.gaussianish((T, N), seed + 3))
    last = synth.gaussianish((N,), seed + 4).to(DEV)
    vr = RunningMeanStd((1,), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    stats = torch.zeros(8, device=DEV)
    if static:
        print(f'mrl_gae_finish_static N={N} T={T}', flush=True)
        _hip.check(L.mrl_gae_finish_static(buf.c, _hip.ptr(last), 0.99, 0.95, vr.c_struct(), _hip.ptr(stats), None), 'static')
    else:
        ws = torch.empty(((N + 63) // 64) * 48 + 256, dtype=torch.uint8, device=DEV)
        _hip.check(L.mrl_gae_finish(buf.c, _hip.ptr(last), 0.99, 0.95, vr.c_struct(), _hip.ptr(stats), _hip.ptr(ws), ws.numel(), None), 'ws')
    torch.cuda.synchronize()
    out = {k: buf._flat[k].cpu().clone() for k in ('returns', 'values', 'advantages', 'neglogp')}
    out['stats'] = stats.cpu()[:6].clone()
    out.update({k: v.cpu().clone() for k, v in vr.state_dict().items()})
    return out


def _assert_allocates_nothing(N, T, seed):
    """The device's free memory before and after a call of the entry point with every argument prepared beforehand (the first
    call on a device, which looks up the scratch array's address and may load the code object, has happened: callers that
    capture graphs make one call outside the capture for the same reason)."""
    from mineral_amd import _hip
    from mineral_amd.agents.ppo.experience import ExperienceBuffer
    from mineral_amd.common.normalizers import RunningMeanStd
    L = _hip.lib()
    buf = ExperienceBuffer(N, T, N * T, N * T, {'obs': (8,)}, 2, DEV)
    s = buf.storage_dict
    s['rewards'].copy_(synth.gaussianish((T, N, 1), seed))
    s['values'].copy_(synth.gaussianish((T, N, 1), seed + 1))
    last = synth.gaussianish((N,), seed + 4).to(DEV)
    vr = RunningMeanStd((1,), eps=1e-5, with_clamp=True, initial_count=1, device=DEV)
    stats = torch.zeros(8, device=DEV)
    c, vc = buf.c, vr.c_struct()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    _hip.check(L.mrl_gae_finish_static(c, _hip.ptr(last), 0.99, 0.95, vc, _hip.ptr(stats), None), 'static')
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, 'the entry point must not allocate device memory'


def test_static_scratch_entry_point_matches_the_workspace_one():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    # the larger rollout first, then a smaller one: the scratch still holds the partial sums of the first call
    for N, T, seed in ((70001, 16, 30), (333, 7, 40)):
        got = _run(N, T, True, seed)
        if seed == 40:
            _assert_allocates_nothing(N, T, seed)
        ref = _run(N, T, False, seed)
        for k in ref:
            assert torch.equal(got[k], ref[k]), f'{k} differs at N={N}, T={T}'
        assert torch.isfinite(got['advantages']).all()


def test_static_scratch_entry_point_rejects_sizes_beyond_its_maximum():
    from mineral_amd import _hip
    L = _hip.lib()
    c = _hip.Rollout()
    c.num_envs, c.horizon = (1 << 20) + 1, 4
    rc = L.mrl_gae_finish_static(c, None, 0.99, 0.95, None, None, None)
    assert rc == -1 and b'at most 1048576 envs' in L.mrl_last_error()
