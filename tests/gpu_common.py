"""Plumbing shared by the `-m gpu` tests: build the HIP-backed render_fn for a scene and
load the regenerated numpy weights into it by the reference's state_dict keys."""
import numpy as np
import torch

from hyperreel_amd.render import build_render_fn


def to_torch_state_dict(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def make_render_fn(cfg, dataset, sd, device='cuda', mlp_precision='auto', grid_dtype='fp32', iteration=None):
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    fn = build_render_fn(cfg, dataset=dataset, grid_size=grid, device=device, mlp_precision=mlp_precision, grid_dtype=grid_dtype)
    missing, unexpected = fn.model.load_state_dict(to_torch_state_dict(sd), strict=False)
    # the synthetic scenes carry no dummy_layer entries; everything else must be present
    assert not [m for m in missing if 'dummy_layer' not in m], missing
    assert not unexpected, unexpected
    if iteration is not None:                 # inside the activation / encoding warm-up windows (INRSystem.set_train_iter)
        fn.model.set_iter(iteration)
    return fn


def render_np(fn, rays, want=()):
    r = torch.from_numpy(np.ascontiguousarray(rays, np.float32)).cuda()
    out = fn.model.render(r, want=want)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}
