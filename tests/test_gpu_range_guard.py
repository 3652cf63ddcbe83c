"""The fp16 range guard of the MLP arithmetic (include/hyperreel_hip.h: HR_MLP_AUTO, HR_E_RANGE, hr_model_calibrate,
HR_OPT_MLP_OVERFLOW).  The reference's BaseMLP (nlf/nets/mlp.py:127-172) is fp32: any finite activation is legal there, so a
model whose activations leave the IEEE-half range must still render correctly in the default mode and must never render
infinities silently in a forced fp16 mode."""
import numpy as np
import pytest
import torch

from helpers import Golden, linf

pytestmark = pytest.mark.gpu

L = 'model.embedding_model.embeddings.0.net.layers.'


def scaled_network(sd, log2_s):
    """The SAME function with layer 0's outputs larger by 2^log2_s: LeakyReLU is positively homogeneous, so (W0, b0) * s followed
    by W1 / s is the network it was -- exactly, in fp32, for a power of two -- while the activations between them are s times
    as large."""
    s = np.float32(2.0 ** log2_s)
    out = dict(sd)
    out[L + '0.0.weight'] = sd[L + '0.0.weight'] * s
    out[L + '0.0.bias'] = sd[L + '0.0.bias'] * s
    out[L + '1.0.weight'] = sd[L + '1.0.weight'] / s
    return out


AUTO = 'f16f8'       # what 'auto' resolves to on the shipped families since round 5: the verified fast path (HR_MLP_F16F8V), fp16 halves like f16x3


def test_auto_resolves_to_the_verified_fp16_path_on_the_shipped_families_and_no_ray_overflows():
    from gpu_common import make_render_fn, render_np
    for case in ('donerf_sphere_small', 'technicolor_z_plane_small'):
        g = Golden(case)
        fn = make_render_fn(g.cfg, g.dataset, g.state_dict)
        assert fn.model.mlp_precision_active() == AUTO and fn.model.mlp_verified()
        got = render_np(fn, g.rays)['rgb']
        assert linf(got, g.rgb) <= 1e-4
        assert not fn.model.mlp_overflowed()
        # the same decision on real rays, with the measured range: O(1-100) for these MLPs, three orders below the limit
        amax = fn.model.calibrate(torch.from_numpy(g.rays).cuda())
        assert fn.model.mlp_precision_active() == AUTO and fn.model.mlp_verified()
        assert len(amax) == len([k for k in g.state_dict if k.startswith(L) and k.endswith('weight')])
        assert 0.0 < max(amax) < 65504.0 / 8.0, amax


def test_large_activations_fall_back_to_bf16x3_and_still_match_the_reference():
    """activations ~2e5 between layers 0 and 1: `auto` must pick the fp32-range split and match the reference's render of the
    (identical) network to the north-star tolerance"""
    from gpu_common import make_render_fn, render_np
    g = Golden('donerf_sphere_small')
    sd = scaled_network(g.state_dict, 17)
    fn = make_render_fn(g.cfg, g.dataset, sd)
    assert fn.model.mlp_precision_active() == 'bf16x3'
    got = render_np(fn, g.rays)['rgb']
    assert linf(got, g.rgb) <= 1e-4
    assert not fn.model.mlp_overflowed()                     # bf16 halves: nothing to overflow
    amax = fn.model.calibrate(torch.from_numpy(g.rays).cuda())
    assert amax[1] > 65504.0 and fn.model.mlp_precision_active() == 'bf16x3', amax
    exact = render_np(make_render_fn(g.cfg, g.dataset, sd, mlp_precision='fp32'), g.rays)['rgb']
    assert linf(got, exact) <= 5e-5


@pytest.mark.parametrize('forced', ['f16x3', 'f16x2', 'f16f8'])
def test_forced_fp16_arithmetic_is_refused_by_name_when_it_would_overflow(forced):
    from gpu_common import make_render_fn
    from hyperreel_amd.lib import HipRangeError
    g = Golden('donerf_sphere_small')
    with pytest.raises(HipRangeError, match='65504'):
        make_render_fn(g.cfg, g.dataset, scaled_network(g.state_dict, 17), mlp_precision=forced).model.native()
    # ... and accepted where it fits (2^6: activations of a few hundred)
    fn = make_render_fn(g.cfg, g.dataset, scaled_network(g.state_dict, 6), mlp_precision=forced)
    assert fn.model.mlp_precision_active() == forced


def test_sticky_overflow_bit_reports_rendered_rays_that_leave_the_half_range():
    """calibration cannot see every ray: rays whose Pluecker moment is ~1e6 (origins a million scene units away) put input features
    beyond 65504 -- a forced fp16 arithmetic must say so (HR_OPT_MLP_OVERFLOW), on both execution plans; the default (the verified path)
    repairs those rays on the device instead (f16f8 -> f16x3 -> bf16x3 tiles, tests/test_gpu_verified.py) and has nothing to report"""
    from gpu_common import make_render_fn, render_np
    g = Golden('donerf_sphere_small')
    far = g.rays.copy()
    far[:, :3] *= 1e6
    for frame_kernel in (True, False):
        fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision='f16x3')
        fn.model.set_execution(frame_kernel=frame_kernel)
        render_np(fn, g.rays)
        assert not fn.model.mlp_overflowed()
        fn.model._render_calls = 16           # past the calls on which render() polls the bit itself (and would refuse: the test below)
        render_np(fn, far)
        assert fn.model.mlp_overflowed()
        fn.model.calibrate(torch.from_numpy(g.rays).cuda())      # a new calibration clears the bit
        assert not fn.model.mlp_overflowed()
        with pytest.raises(Exception, match='bf16x3'):
            # forcing f16x3 on THOSE rays is refused
            make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision='f16x3').model.calibrate(torch.from_numpy(far).cuda())
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict)
    assert fn.model.mlp_precision_active() == AUTO
    img = render_np(fn, far)['rgb']
    ref = render_np(make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision='bf16x3'), far)['rgb']
    assert np.isfinite(img).all() and np.array_equal(img, ref) and fn.model.wide_count() >= far.shape[0]
    assert not fn.model.mlp_overflowed() and fn.model.mlp_precision_active() == AUTO
    fn.model.calibrate(torch.from_numpy(far).cuda())             # `auto` calibrated on THOSE rays takes the fp32-range split outright
    assert fn.model.mlp_precision_active() == 'bf16x3'


def test_first_render_call_checks_the_bit_and_falls_back_to_bf16x3():
    """the arithmetic is chosen on synthetic calibration rays; a model whose FIRST real batch leaves the half range must not
    return an image made from saturated operands.  The verified default repairs such rays on the device up to its third list's capacity
    (8192 rays); a batch with more of them fills the list, the kernels raise the sticky bit, and render() -- which reads it on the first
    calls -- re-decides on those rays (auto -> bf16x3) and renders the batch again"""
    from gpu_common import make_render_fn, render_np
    g = Golden('donerf_sphere_small')
    far = np.concatenate([g.rays] * 80, 0)
    far[:, :3] *= 1e6
    assert far.shape[0] > 2 * 8192
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict)
    assert fn.model.mlp_precision_active() == AUTO
    with pytest.warns(UserWarning, match='IEEE-half range'):
        img = render_np(fn, far)['rgb']
    assert fn.model.mlp_precision_active() == 'bf16x3' and not fn.model.mlp_overflowed()
    ref = render_np(make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision='bf16x3'), far)['rgb']
    assert np.isfinite(img).all() and np.array_equal(img, ref)
    # a forced fp16 mode is refused loudly instead
    from hyperreel_amd.lib import HipRangeError
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision='f16x3')
    with pytest.warns(UserWarning), pytest.raises(HipRangeError):
        render_np(fn, far)
