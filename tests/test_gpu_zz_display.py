"""`-m gpu`: device runs of hr_pack_display, a checkpoint with a shrunk box, hr_dense_alpha / updateAlphaMask / shrink
and the forward-mode intersection gradients.  Their arithmetic is also checked on the CPU through the host builds of the
same sources (tests/test_host_math.py, tests/test_alpha_mask_host.py).  Ordinary tests: they passed on the round-1
driver run (GPUTEST_r01.json) and gate the suite like every other file."""
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


@pytest.mark.parametrize('case', ['alpha_mask_static', 'alpha_mask_video', 'alpha_mask_video_open'])
def test_occupancy_mask_and_shrink_follow_the_reference(case):
    """TensorBase.set_iter at an update_AlphaMask_list iteration (tensorf_base.py:510-530): hr_dense_alpha -> max-pool ->
    threshold -> shrink, against what the reference's own code produced (tests/golden/mask, oracle/refgen/make_alpha_mask.py),
    and the model still renders afterwards."""
    import json
    import os
    from hyperreel_amd import config as cfgmod
    from hyperreel_amd import scenes
    from gpu_common import make_render_fn
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mask', case + '.npz'))
    r = json.loads(bytes(z['recipe']).decode())
    cfg, ds = cfgmod.model_config(r['model']), r['dataset']
    if 'thre' in r:                         # 'alpha_mask_video_open': holes inside the mask's box, where the density is not zero --
        cfg['color']['net']['alpha_mask_thre'] = r['thre']     # the case in which a keyframe net must NOT consult the previous mask
    sd = scenes.make_state_dict(cfg, ds, r['grid'], r['seed'], 'dense', 1.0)
    if r.get('carve', True):
        sd = scenes.carve_density(sd)
    fn = make_render_fn(cfg, ds, sd)
    net = fn.model.color_model.net
    a1 = net.getDenseAlpha(r['n1'])
    assert float((a1.cpu() - torch.from_numpy(z['alpha1'])).abs().max()) <= 1e-6
    new_aabb = net.updateAlphaMask(r['n1'])
    assert np.array_equal(net.alpha_volume.cpu().numpy(), z['mask_volume'])
    net.shrink(new_aabb)
    assert np.abs(net.aabb.cpu().numpy() - z['aabb_after']).max() <= 1e-6 and fn.model.grid_size == z['grid_after'].tolist()
    a2 = net.getDenseAlpha(r['n2'])
    assert float((a2.cpu() - torch.from_numpy(z['alpha2'])).abs().max()) <= 1e-6
    video = cfg.color.net.type == 'tensor_vm_split_time'
    rays = torch.from_numpy(scenes.random_rays(64, 3, video)).cuda()
    assert bool(torch.isfinite(fn.model.render(rays)['rgb']).all())


@pytest.mark.parametrize('case', ['sweep/technicolor_cascaded', 'sweep/shiny_z_plane_cascaded', 'sweep/bom_sphere', 'sweep/shiny_z_deformable'])
def test_training_gradients_of_cascades_and_long_intersections(case):
    """The models whose derivative was completed after the GPU budget was spent: point_prediction cascades (coarse rows
    stage + fine sample stage, both MLPs in autograd) and the forward-mode intersections; every trainable tensor against
    torch.autograd on the CPU restatement, as tests/test_gpu_train.py does for the others."""
    from gpu_common import make_render_fn
    from torch_port import TorchPort
    g = Golden(case)
    n = min(96, g.rays.shape[0])
    rays = np.ascontiguousarray(g.rays[:n], np.float32)
    G = np.random.default_rng(3).standard_normal((n, 3)).astype(np.float32)
    port = TorchPort(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    grids = [t.requires_grad_(True) for grp in (port.d_a, port.d_b, port.a_a, port.a_b) for t in grp]
    port.basis.requires_grad_(True)
    rgb_ref = port.color(port.embed(torch.from_numpy(rays)), train=True, white_bg=False)
    (rgb_ref * torch.from_numpy(G)).sum().backward()
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    fn.train()
    rgb = fn.model.forward_train(torch.from_numpy(rays).cuda(), white_bg=False)
    (rgb * torch.from_numpy(G).cuda()).sum().backward()
    assert float((rgb.detach().cpu() - rgb_ref.detach()).abs().max()) <= 2e-5
    from hyperreel_amd.train import grid_parameters
    vm = fn.model.color_model.net
    for p, ref in zip(grid_parameters(vm) + [vm.basis_mat.weight], grids + [port.basis]):
        if p.numel() == 0 or ref.grad is None or float(ref.grad.abs().max()) == 0:
            continue
        assert float((p.grad.cpu().reshape(ref.grad.shape) - ref.grad).abs().max()) <= 1e-3 * float(ref.grad.abs().max()) + 1e-7
    # the MLPs: the gradient reached them (their values are autograd's own once d_head is right, which the planes certify)
    got = [p.grad for m in fn.model.embedding_model.embeddings if hasattr(m, 'net') and hasattr(m.net, 'layers') for p in m.net.parameters()]
    assert got and all(gp is not None and bool(torch.isfinite(gp).all()) for gp in got)


def test_occupancy_early_reject_equals_the_reference_with_its_mask_enabled():
    """Opt-in occupancy test of the render path (hr_model_set_occupancy): the image must equal what the REFERENCE renders
    when the test it ships disabled (`if self.alphaMask is not None and False`, tensorf_no_sample.py:171) is switched on --
    tests/golden/mask/alpha_mask_render.npz, made by oracle/refgen/make_alpha_mask.py from the reference's own forward
    source on an uncarved scene whose mask threshold cuts into real density (masked vs shipped image: L-inf 0.69).
    Off, the image is the shipped one; both execution plans agree bit for bit."""
    import json
    import os
    from gpu_common import make_render_fn, to_torch_state_dict
    from helpers import GOLDEN_DIR
    from hyperreel_amd import config as C, scenes
    z = np.load(os.path.join(GOLDEN_DIR, 'mask', 'alpha_mask_render.npz'))
    r = json.loads(bytes(z['recipe']).decode())
    cfg = C.model_config(r['model'])
    cfg['color']['net']['alpha_mask_thre'] = r['thre']
    sd = scenes.make_state_dict(cfg, r['dataset'], r['grid'], r['seed'], 'dense', 1.0)
    fn = make_render_fn(cfg, r['dataset'], sd, mlp_precision='f16x3')     # (both execution plans below: 'auto' -- the verified two-pass plan -- has only one)
    net = fn.model.color_model.net
    rays = torch.from_numpy(np.ascontiguousarray(z['rays'], np.float32)).cuda()
    plain = fn.model.render(rays)['rgb'].cpu().numpy()
    assert np.abs(plain - z['rgb_plain']).max() <= 1e-4
    net.updateAlphaMask(tuple(r['n1']))                                  # the mask, built on the device (hr_dense_alpha)
    assert np.array_equal(net.alpha_volume.cpu().numpy(), z['mask_volume'])
    fn.model.set_occupancy(True)
    masked = fn.model.render(rays)['rgb']
    assert np.abs(masked.cpu().numpy() - z['rgb_masked']).max() <= 1e-4
    assert np.abs(z['rgb_masked'] - z['rgb_plain']).max() > 0.1          # the fixture distinguishes the two behaviours
    # both execution plans reject the same samples with the same arithmetic: bit-identical images
    fn.model.set_execution(frame_kernel=True)
    assert fn.model.frame_kernel_active()
    one = fn.model.render(rays)['rgb'].clone()
    fn.model.set_execution(frame_kernel=False)
    assert not fn.model.frame_kernel_active()
    two = fn.model.render(rays)['rgb'].clone()
    assert torch.equal(one, two), f'{int((one != two).any(-1).sum())} rays differ between the plans'
    assert np.abs(two.cpu().numpy() - z['rgb_masked']).max() <= 1e-4
    fn.model.set_occupancy(False)
    assert np.abs(fn.model.render(rays)['rgb'].cpu().numpy() - z['rgb_plain']).max() <= 1e-4
