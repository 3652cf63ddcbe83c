"""Shared test plumbing: golden fixtures -> (cfg, dataset scalars, regenerated weights, rays, expected)."""
import glob
import json
import os

import numpy as np

from hyperreel_amd import config as C
from hyperreel_amd import scenes

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))


def sweep_cases():
    """One fixture per shipped model YAML the backend accepts (oracle/refgen/make_sweep.py)."""
    return sorted('sweep/' + os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, 'sweep', '*.npz')))


def sweep_coverage():
    with open(os.path.join(GOLDEN_DIR, 'sweep', 'coverage.json')) as f:
        return json.load(f)


class Golden:
    def __init__(self, case):
        z = np.load(os.path.join(GOLDEN_DIR, case + '.npz'))
        self.arrays = {k: z[k] for k in z.files if k != 'recipe'}
        self.recipe = json.loads(bytes(z['recipe']).decode())
        r = self.recipe
        if 'model_cfg' in r:        # sweep fixtures carry the parsed YAML group (no /root/reference on the GPU box)
            self.cfg = C.epoch_to_iter(C.to_cfg(r['model_cfg']), 4000)
        else:
            self.cfg = C.model_config(r['model'], z_channels=r['z_channels'])
        self.iteration = r.get('iter')          # training iteration of the activation / PE schedules (None: converged)
        self.dataset = r['dataset']
        self.grid = r['grid']
        self.rays = self.arrays['rays']
        if 'frame_idx' in self.arrays:          # rays of the 800x800 benchmark frame, stored as pixel indices (make_golden.py)
            at, frame = (int(v) for v in self.arrays['frame_at'])
            part = scenes.benchmark_rays(r['model'], 800, 800, frame=frame)[self.arrays['frame_idx']]
            self.rays = np.ascontiguousarray(np.concatenate([self.rays[:at], part, self.rays[at:]], 0), np.float32)
        self.rgb = self.arrays['rgb']
        self._sd = None

    @property
    def state_dict(self):
        if self._sd is None:
            r = self.recipe
            self._sd = scenes.make_state_dict(self.cfg, self.dataset, r['grid'], r['seed'], r['density'], r['app_scale'])
            got = scenes.state_dict_checksum(self._sd)
            assert abs(got - r['checksum']) <= 1e-6 * max(1.0, abs(r['checksum'])), \
                f'regenerated weights differ from the ones the golden was made with ({got} vs {r["checksum"]})'
        return self._sd


def linf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def trainable_sweep_cases(cascades=False):
    """Sweep fixtures whose model the training path differentiates (mirror of hr_train_unsupported, csrc/hr_train.h) --
    which are also the ones oracle/torch_port.py restates.  cascades: the point_prediction models instead of the
    single-level ones (their coarse and fine stages are checked separately)."""
    from hyperreel_amd import plan
    out = []
    for c in sweep_cases():
        g = Golden(c)
        if plan.is_cascade(g.cfg) != bool(cascades):
            continue
        with plan.at_iteration(g.iteration):
            hc = plan.compile_model(g.cfg, g.dataset, g.grid, iteration=g.iteration)[1]
        out.append(c)
    return out
