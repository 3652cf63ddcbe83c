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


def initialiser_golden_cases():
    """golden_cases() without the fixtures whose MLP is scaled away from the initialiser (scenes.MLP_VARIANTS): the reduced-precision
    arithmetics that are opt-in by name (f16x2, plain f16f8, bf16x3) carry an error proportional to the weights and are not held to the
    1e-4 bar there -- 'auto' (which is), f16x3 and fp32 are."""
    return [c for c in golden_cases() if not c.endswith(('_hostile', '_stiff'))]


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
            self._sd = scenes.make_state_dict(self.cfg, self.dataset, r['grid'], r['seed'], r['density'], r['app_scale'], r.get('mlp', 'default'))
            got = scenes.state_dict_checksum(self._sd)
            assert abs(got - r['checksum']) <= 1e-6 * max(1.0, abs(r['checksum'])), \
                f'regenerated weights differ from the ones the golden was made with ({got} vs {r["checksum"]})'
            for k in self.arrays:              # tensors the fixture stores in full (post-fit weights: oracle/refgen/make_golden.py, POSTFIT)
                if k.startswith('sd/'):
                    self._sd[k[3:]] = self.arrays[k]
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


class GradGolden:
    """Gradients of sum(rgb * G) from the REFERENCE'S OWN autograd in train mode (oracle/refgen/make_grad_golden.py): full arrays for
    small tensors, 16 seeded projections + the L2 norm for the MLP's large matrices."""

    def __init__(self, case, white):
        z = np.load(os.path.join(GOLDEN_DIR, 'grad', f'{case}_bg{int(white)}.npz'))
        self.recipe = json.loads(bytes(z['recipe']).decode())
        self.rgb = z['rgb']
        self.full = {k[5:]: z[k] for k in z.files if k.startswith('full/')}
        self.proj = {k[5:]: z[k] for k in z.files if k.startswith('proj/')}
        self.norm = {k[5:]: z[k] for k in z.files if k.startswith('norm/')}
        self.n_rays = int(self.recipe['n_rays'])
        self.G = np.random.default_rng(self.recipe['g_seed']).standard_normal((self.n_rays, 3)).astype(np.float32)

    def names(self):
        return sorted(set(self.full) | set(self.proj))

    def check(self, name, got, rel):
        """got: the gradient of tensor `name` (any shape with the same element order).  rel: tolerance relative to the reference
        gradient's largest element (full) / to its norm (projections: each is a N(0, |g|^2) combination of the elements)."""
        got = np.asarray(got, np.float64).ravel()
        if name in self.full:
            want = self.full[name].astype(np.float64).ravel()
            assert got.size == want.size, (name, got.size, want.size)
            scale = np.abs(want).max()
            err = np.abs(got - want).max()
            assert err <= rel * scale + 1e-9, f'{name}: |err| {err:.3e} vs max |g| {scale:.3e}'
        else:
            norm, size = self.norm[name]
            assert got.size == int(size), (name, got.size, size)
            seed = (sum(ord(c) * (i + 1) for i, c in enumerate(name)) * 2654435761 + got.size) % (2 ** 32)
            v = np.random.default_rng(seed).standard_normal((16, got.size)).astype(np.float32).astype(np.float64)
            err = np.abs(v @ got - self.proj[name]).max()
            assert err <= rel * norm * 4.0 + 1e-9, f'{name}: projection error {err:.3e} vs |g| {norm:.3e}'
            assert abs(np.linalg.norm(got) - norm) <= rel * norm + 1e-9, name


def port_leaves(port):
    """TorchPort tensors under the reference's parameter names (the keys of a gradient golden)."""
    out = {}
    net = 'model.color_model.net.'
    video = bool(port.o.video) if hasattr(port, 'o') else bool(getattr(port, 'video', False))
    a_name, b_name = ('plane_space', 'plane_time') if video else ('plane', 'line')
    for kind, ga, gb in (('density', port.d_a, port.d_b), ('app', port.a_a, port.a_b)):
        for j in range(3):
            out[f'{net}{kind}_{a_name}.{j}'] = ga[j]
            out[f'{net}{kind}_{b_name}.{j}'] = gb[j]
    out[net + 'basis_mat.weight'] = port.basis
    n = len(port.layers)
    for i, (w, b) in enumerate(port.layers):
        mid = f'{i}.0' if i < n - 1 else f'{i}'
        out[f'model.embedding_model.embeddings.0.net.layers.{mid}.weight'] = w
        out[f'model.embedding_model.embeddings.0.net.layers.{mid}.bias'] = b
    return out


def build_host_lib(out, src, deps):
    """g++ -shared of a host restatement, safe under pytest-xdist: one builder at a time (flock), the library appears atomically."""
    import fcntl, subprocess
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out + '.lock', 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            tmp = f'{out}.{os.getpid()}.tmp'
            subprocess.run(['g++', '-O1', '-ffp-contract=off', '-fno-fast-math', '-shared', '-fPIC', '-o', tmp, src], check=True)
            os.replace(tmp, out)
    return out
