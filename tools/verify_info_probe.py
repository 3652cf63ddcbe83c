"""Prints hr_model_verify_info (the verified fast path's per-model band and what it was measured from) for the benchmark families,
the small goldens and the hostile fixtures.  python tools/verify_info_probe.py [--full]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_common import make_render_fn  # noqa: E402
from helpers import Golden, golden_cases  # noqa: E402
from hyperreel_amd import config as C, scenes  # noqa: E402


def show(tag, m):
    m.native()
    v = m.verify_info()
    print(f"{tag:34s} ver {v['verified']} fb {v['fallback']} band {v['band']:.2e} q {v['band_q']:.2e} off {v['band_off']:.2e} | d_zc {v['max_d_zc']:.2e} d_dn {v['max_d_dist_n']:.2e} "
          f"d_geo {v['max_d_geo_n']:.2e} d_off {v['max_d_off']:.2e} d_dist {v['max_d_dist']:.2e} d_head {v['max_d_head']:.2e} d_rgb {v['max_d_rgb']:.2e} listed {v['listed_frac']:.4f} "
          f"rays {v['n_rays_used']}/{v['n_rays']} flip {v['n_flipped']} shaky {v['n_shaky']} n {v['n_samples']}", flush=True)


for case in golden_cases():
    g = Golden(case)
    if '--full' not in sys.argv and max(g.recipe['grid']) > 128:
        continue
    m = make_render_fn(g.cfg, g.dataset, g.state_dict, iteration=g.iteration).model
    show(case, m)
    m.calibrate(torch.from_numpy(g.rays).cuda())
    show(case + ' @rays', m)
for model in ('donerf_sphere', 'technicolor_z_plane', 'immersive_sphere', 'neural_3d_z_plane'):
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    sd = scenes.make_state_dict(cfg, ds, [64, 64, 64] if '--full' not in sys.argv else None, seed=7, density='dense', app_scale=1.0)
    m = make_render_fn(cfg, ds, sd).model
    show(model, m)
    m.calibrate(torch.from_numpy(scenes.benchmark_rays(model, 800, 800, frame=7)).cuda())
    show(model + ' @frame', m)
