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
