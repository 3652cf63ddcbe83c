"""Gradient goldens from the REFERENCE'S OWN autograd: the reference modules (imported through
ref_shim.py) in TRAIN mode with requires_grad, loss = sum(rgb * G) for a seeded G, both background draws
(`white_bg or (training and rand() < 0.5)`, tensorf_no_sample.py:236 -- torch.rand is pinned for the call).

    python oracle/refgen/make_grad_golden.py            ->  tests/golden/grad/<case>_bg<0|1>.npz

Per trainable tensor of the path (state_dict key): the full gradient when it has <= 20 000 elements, otherwise its L2 norm
and 16 seeded random projections (the MLP's 256x256 matrices: 1.6 MB per scene in full).  The fixtures pin (a) the CPU
restatement's autograd (tests/test_oracle_golden.py), which the host-build gradient checks compare against, and (b) the
HIP training step itself (tests/test_gpu_train.py).  TEST INFRASTRUCTURE ONLY."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_shim  # noqa: E402
from helpers import Golden  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'grad')
CASES = ['donerf_sphere_small', 'donerf_cylinder_small', 'technicolor_z_plane_small', 'neural_3d_z_plane_small', 'immersive_sphere_small']
N_RAYS = 192
FULL_MAX = 20000
N_PROJ = 16


def projections(name, g):
    """16 seeded random projections of a flat gradient (the seed depends on the tensor's name and size only)."""
    seed = (sum(ord(c) * (i + 1) for i, c in enumerate(name)) * 2654435761 + g.size) % (2 ** 32)
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((N_PROJ, g.size)).astype(np.float32)
    return (v.astype(np.float64) @ g.astype(np.float64).ravel()).astype(np.float64)


def main():
    os.makedirs(OUT, exist_ok=True)
    for case in CASES:
        g = Golden(case)
        r = g.recipe

        def overrides(cfg):
            if r['z_channels'] is not None:
                cfg.embedding.embeddings.ray_prediction_0.z_channels = r['z_channels']
                cfg.embedding.embeddings.ray_intersect_0.z_channels = r['z_channels']
            cfg.color.net.grid_size = ref_shim.to_attr({'start': list(r['grid']), 'end': list(r['grid'])})
        fn = ref_shim.build_reference(ref_shim.load_model_cfg(r['model'], overrides), g.dataset)
        own = dict(fn.state_dict())
        with torch.no_grad():
            for k, v in g.state_dict.items():
                if not k.endswith('gridSize'):
                    own[k].copy_(torch.from_numpy(v))
        fn.train()
        n = min(N_RAYS, g.rays.shape[0])
        rays = torch.from_numpy(np.ascontiguousarray(g.rays[:n], np.float32))
        G = torch.from_numpy(np.random.default_rng(3).standard_normal((n, 3)).astype(np.float32))
        for white in (0, 1):
            for p in fn.parameters():
                p.grad = None
            real_rand = torch.rand
            torch.rand = lambda *a, **k: torch.full((1,), 0.1 if white else 0.9)      # the background draw of this step
            try:
                with ref_shim.cpu_mode():
                    rgb = fn(rays)['rgb']
            finally:
                torch.rand = real_rand
            (rgb * G).sum().backward()
            payload = {'rgb': rgb.detach().numpy().astype(np.float32),
                       'recipe': np.frombuffer(json.dumps({'case': case, 'n_rays': n, 'white': white, 'g_seed': 3}).encode(), dtype=np.uint8)}
            n_full = n_proj = 0
            for name, p in fn.named_parameters():
                if p.grad is None or 'dummy' in name:
                    continue
                gr = p.grad.detach().numpy().astype(np.float32)
                if gr.size <= FULL_MAX:
                    payload['full/' + name] = gr
                    n_full += 1
                else:
                    payload['proj/' + name] = projections(name, gr)
                    payload['norm/' + name] = np.asarray([np.linalg.norm(gr.astype(np.float64)), float(gr.size)])
                    n_proj += 1
            path = os.path.join(OUT, f'{case}_bg{white}.npz')
            np.savez_compressed(path, **payload)
            print(f'{case} bg{white}: {n_full} full + {n_proj} projected gradients, {os.path.getsize(path) / 1024:.0f} KiB', flush=True)


if __name__ == '__main__':
    main()
