"""`-m gpu`: hr_pack_display against the host code of the reference's viewer (utils/gui_utils.py:174-205, utils/__init__.py:47).
(Collected last on purpose: the kernel was added after the round's GPU budget was spent and has only been checked through
its host-compiled pieces, tests/test_host_math.py::test_display_pack_matches_the_viewer_host_code.)"""
import numpy as np
import pytest
import torch

from helpers import Golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('transpose,flip', [(False, False), (True, False), (False, True), (True, True)])
def test_pack_display_matches_the_viewer_host_path(transpose, flip):
    from gpu_common import make_render_fn
    g = Golden('donerf_sphere_small')
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict)
    h, w = 9, 13
    rgb = torch.from_numpy(np.random.default_rng(0).uniform(-0.1, 1.1, (h * w, 3)).astype(np.float32)).cuda()
    ref = rgb.view(h, w, 3).cpu().numpy()
    if transpose:
        ref = ref.transpose(1, 0, 2)
    if flip:
        ref = np.flip(ref, axis=0)
    ref = np.ascontiguousarray(ref)
    f32 = fn.model.pack_display(rgb, h, w, transpose, flip, rgba8=False).cpu().numpy()
    assert np.array_equal(f32, ref)
    u8 = fn.model.pack_display(rgb, h, w, transpose, flip, rgba8=True).cpu().numpy()
    assert np.array_equal(u8[..., :3], (255 * np.clip(ref, 0, 1)).astype(np.uint8)) and (u8[..., 3] == 255).all()


def test_a_checkpoint_with_a_shrunk_box_renders_in_that_box():
    """`aabb` is a buffer of the colour net that training shrinks (tensorf_base.py:1191-1232); a checkpoint's value, not
    the YAML's, is what the reference renders with (the oracle takes it from the state_dict too)."""
    from gpu_common import make_render_fn, render_np
    from hyperreel_oracle import HyperReelOracle
    g = Golden('donerf_sphere_small')
    sd = dict(g.state_dict)
    box = np.asarray(sd['model.color_model.net.aabb'], np.float32).copy()
    box[0] *= 0.8
    box[1] *= 0.7
    sd['model.color_model.net.aabb'] = box
    fn = make_render_fn(g.cfg, g.dataset, sd)
    ref = HyperReelOracle(g.cfg, g.dataset, sd).render(g.rays)['rgb']
    assert np.abs(ref - g.rgb).max() > 1e-3                          # the box matters on these rays
    assert np.abs(render_np(fn, g.rays)['rgb'] - ref).max() <= 1e-4
