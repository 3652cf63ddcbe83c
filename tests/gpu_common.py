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


# ---- the poison kernel (tests/c_abi/poison.hip): test infrastructure, built in-tree by __graft_entry__.build() so that it travels
POISON_SRC = __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'c_abi', 'poison.hip')
POISON_LIB = __import__('os').path.join(__import__('os').path.dirname(POISON_SRC), '_build', 'libhr_poison.so')


def build_poison(force=False):
    """hipcc --offload-arch=gfx950 tests/c_abi/poison.hip -> tests/c_abi/_build/libhr_poison.so (cross-compiles without a GPU)."""
    import os
    import subprocess
    from hyperreel_amd.build import hipcc
    if not force and os.path.exists(POISON_LIB) and os.path.getmtime(POISON_LIB) >= os.path.getmtime(POISON_SRC):
        return POISON_LIB
    os.makedirs(os.path.dirname(POISON_LIB), exist_ok=True)
    subprocess.run([hipcc(), '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', POISON_SRC, '-o', POISON_LIB], check=True)
    return POISON_LIB


class Poison:
    """poison(pattern): every VGPR and LDS word of every CU holds `pattern` when the next kernel of the current stream starts."""

    def __init__(self):
        import ctypes
        self.lib = ctypes.CDLL(build_poison())
        self.lib.hr_poison.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.hr_poison.restype = ctypes.c_int
        self.touched = torch.zeros(1, dtype=torch.int32, device='cuda')

    def __call__(self, pattern):
        rc = self.lib.hr_poison(int(pattern) & 0xFFFFFFFF, self.touched.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, f'hr_poison: {rc}'
