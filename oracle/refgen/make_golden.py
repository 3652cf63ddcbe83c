"""Generates tests/golden/*.npz by running the REFERENCE ITSELF (imported in-process
through ref_shim.py) on seeded synthetic scenes.  Run in the authoring container:

    python oracle/refgen/make_golden.py

TEST INFRASTRUCTURE ONLY.  Each fixture stores the rays, the reference's `rgb`,
the reference's intermediate fields for a subset of rays, the scene recipe
(model name, overrides, dataset scalars, grid size, seed, density variant) and a
checksum of the regenerated weights -- the weights themselves are regenerated
from the seed by `hyperreel_amd.scenes.make_state_dict` on whichever machine
runs the tests, so fixtures stay small.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402
from hyperreel_amd import config as C  # noqa: E402
from hyperreel_amd import scenes  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

# name, model, z override, grid, n_random, pinhole (H,W), density variant, seed, keep intermediates
CASES = [
    dict(case='donerf_sphere_small', model='donerf_sphere', z=None, grid=[44, 36, 28], n_random=160, pin=(8, 10), density='dense', seed=11),
    dict(case='donerf_cylinder_small', model='donerf_cylinder', z=None, grid=[30, 44, 36], n_random=160, pin=(8, 10), density='dense', seed=12),
    dict(case='technicolor_z_plane_small', model='technicolor_z_plane', z=None, grid=[44, 36, 20], n_random=160, pin=(8, 10), density='dense', seed=13),
    dict(case='neural_3d_z_plane_small', model='neural_3d_z_plane', z=None, grid=[40, 30, 26], n_random=96, pin=(6, 8), density='dense', seed=14),
    dict(case='immersive_sphere_small', model='immersive_sphere', z=None, grid=[36, 40, 32], n_random=160, pin=(8, 10), density='dense', seed=15),
    # BASELINE config 1: 4096 random rays, Z=16, tiny 64^3 grid, reference initialiser
    dict(case='config1_random_z16', model='donerf_sphere', z=16, grid=[64, 64, 64], n_random=4096, pin=None, density='default', app_scale=0.1, seed=16, rgb_only=True),
    # full-size shipped grid
    dict(case='donerf_sphere_600', model='donerf_sphere', z=None, grid=[600, 600, 600], n_random=512, pin=(16, 16), density='dense', seed=17, rgb_only=True),
    dict(case='technicolor_full', model='technicolor_z_plane', z=None, grid=[1007, 1007, 503], n_random=256, pin=(16, 16), density='dense', seed=18, rgb_only=True),
    # BASELINE configs[3] / [4] at the shipped final grid: `frame` = seeded subset of the 800x800 benchmark frame
    # (scenes.benchmark_rays, the rays bench.py and the full-frame GPU tests render), so the reference's own pixels
    # sit inside the frame whose >1e-4 rays the GPU test counts
    dict(case='neural_3d_full', model='neural_3d_z_plane', z=None, grid=None, n_random=256, pin=None, frame=(7, 32768), density='dense', seed=7, rgb_only=True),
    dict(case='immersive_full', model='immersive_sphere', z=None, grid=None, n_random=256, pin=None, frame=(7, 32768), density='dense', seed=7, rgb_only=True),
    # HOSTILE sample-prediction MLPs (scenes.MLP_VARIANTS: every hidden Linear x 2, the last x 4, its bias shifted so
    # that the head's activations saturate -- a head 128 times the initialiser's; the comment there says why not more) for the four families
    # at the shipped grid, 32 768 rays of the benchmark frame each: what the verified fast path's per-model band and its fall-back rule
    # (hr_model_finalize) have to hold on -- the f16f8 head error grows with the weights.  `stiff`: the milder variant (head x 15)
    dict(case='donerf_sphere_hostile', model='donerf_sphere', z=None, grid=None, n_random=256, pin=None, frame=(0, 32768), density='dense', seed=31, mlp='hostile', rgb_only=True),
    dict(case='technicolor_hostile', model='technicolor_z_plane', z=None, grid=None, n_random=256, pin=None, frame=(7, 32768), density='dense', seed=32, mlp='hostile', rgb_only=True),
    dict(case='neural_3d_hostile', model='neural_3d_z_plane', z=None, grid=None, n_random=256, pin=None, frame=(7, 32768), density='dense', seed=33, mlp='hostile', rgb_only=True),
    dict(case='immersive_hostile', model='immersive_sphere', z=None, grid=None, n_random=256, pin=None, frame=(7, 32768), density='dense', seed=34, mlp='hostile', rgb_only=True),
    dict(case='donerf_sphere_stiff', model='donerf_sphere', z=None, grid=None, n_random=128, pin=None, frame=(0, 16384), density='dense', seed=35, mlp='stiff', rgb_only=True),
    dict(case='technicolor_stiff', model='technicolor_z_plane', z=None, grid=None, n_random=128, pin=None, frame=(7, 16384), density='dense', seed=36, mlp='stiff', rgb_only=True),
    dict(case='neural_3d_stiff', model='neural_3d_z_plane', z=None, grid=None, n_random=128, pin=None, frame=(7, 16384), density='dense', seed=37, mlp='stiff', rgb_only=True),
    dict(case='immersive_stiff', model='immersive_sphere', z=None, grid=None, n_random=128, pin=None, frame=(7, 16384), density='dense', seed=38, mlp='stiff', rgb_only=True),
]

# POST-FIT weights: the reference's own 200-step Adam run of tests/golden/fit/<case>.npz (make_fit_golden.py, same recipe, same seeds) --
# the trained tensors are stored IN the fixture ('sd/<key>': they are what was trained, not a seed) next to the reference's render of them
POSTFIT = {
    'donerf_sphere_postfit': ('donerf_sphere_fit', 4096, 51),
    'technicolor_postfit': ('technicolor_z_plane_fit', 4096, 52),
}


def special_rays(video, z_plane):
    """Edge cases the reference's arithmetic special-cases."""
    r = [
        [0.0, 0.0, 0.0, 0.0, 0.0, -1.0],            # from the origin, straight down -z
        [0.1, 0.2, 0.9, 1.0, 0.0, 0.0],              # parallel to the z planes: d_z == 0 -> 1e12 divisor
        [0.1, 0.2, 0.9, 0.6, 0.8, 1e-6],             # |d_z| < 1e-5 -> 1e12 divisor
        [10.0, 10.0, 10.0, 0.57735027, 0.57735027, 0.57735027],   # far outside, pointing away
        [5.0, 0.0, 0.0, -1.0, 0.0, 0.0],             # outside the box looking in
        [2.0, 2.0, 2.0, -0.57735027, -0.57735027, -0.57735027],   # exactly on the aabb corner
        [0.0, 0.0, 0.0, 0.0, 1.0, 0.0],              # along the cylinder axis (y): a == 0
        [0.3, -0.1, 0.2, 0.0, 0.0, 1.0],
    ]
    r = np.asarray(r, np.float32)
    if video:
        t = np.asarray([0.0, 1.0, 0.5, 0.020408163, 0.2, 0.98, 0.51, 0.1], np.float32)[:, None]
        r = np.concatenate([r, np.zeros((r.shape[0], 1), np.float32), t], -1)
    return r


def case_rays(c):
    video = c['model'] not in ('donerf_sphere', 'donerf_cylinder')
    z_plane = 'z_plane' in c['model']
    parts = []
    if z_plane:
        parts.append(scenes.random_rays(c['n_random'], c['seed'], video, pos_mean=(0, 0, 1.0), pos_std=0.15,
                                        dir_mean=(0, 0, -1.2), dir_std=0.5))
    else:
        parts.append(scenes.random_rays(c['n_random'], c['seed'], video))
    if c['pin'] is not None:
        H, W = c['pin']
        if z_plane:
            pose = scenes.look_at_pose((0.05, 0.03, 1.0), (0.0, 0.0, -1.0))
        else:
            pose = scenes.look_at_pose((0.3, 0.0, 0.0), (1.0, 0.1, 0.05))
        parts.append(scenes.pinhole_rays(H, W, 40.0, pose, cam_id=0 if video else None,
                                         time=(7.0 / 49.0) if video else None))
    if c.get('frame') is not None:
        frame, n = c['frame']
        full = scenes.benchmark_rays(c['model'], 800, 800, frame=frame)
        idx = np.sort(np.random.default_rng(c['seed']).choice(full.shape[0], n, replace=False))
        c['_frame_at'] = sum(p.shape[0] for p in parts)       # stored as indices, see main()
        c['_frame_idx'] = idx.astype(np.int32)
        parts.append(full[idx])
    parts.append(special_rays(video, z_plane))
    return np.ascontiguousarray(np.concatenate(parts, 0), np.float32)


def build(c):
    model_cfg = C.model_config(c['model'], z_channels=c['z'])
    ds = C.dataset_scalars(c['model'])
    if c['grid'] is None:        # the shipped final grid (what make_state_dict builds without a size)
        from hyperreel_amd.config import n_to_reso
        c['grid'] = [int(v) for v in n_to_reso(model_cfg['color']['net']['N_voxel_final'], model_cfg['color']['net']['aabb'])]
    # reference side: the shipped YAML, plus the same overrides
    def overrides(cfg):
        if c['z'] is not None:
            cfg.embedding.embeddings.ray_prediction_0.z_channels = c['z']
            cfg.embedding.embeddings.ray_intersect_0.z_channels = c['z']
        cfg.color.net.grid_size = ref_shim.to_attr({'start': list(c['grid']), 'end': list(c['grid'])})
    ref_cfg = ref_shim.load_model_cfg(c['model'], overrides)
    fn = ref_shim.build_reference(ref_cfg, ds)
    sd = scenes.make_state_dict(model_cfg, ds, c['grid'], c['seed'], c['density'], c.get('app_scale', 1.0), c.get('mlp', 'default'))
    own = dict(fn.state_dict())
    with torch.no_grad():
        for k, v in sd.items():
            if k.endswith('gridSize'):
                assert own[k].tolist() == v.tolist(), (k, own[k], v)
                continue
            assert tuple(own[k].shape) == tuple(v.shape), (k, own[k].shape, v.shape)
            own[k].copy_(torch.from_numpy(v))
    missing = [k for k in own if k not in sd and 'dummy_layer' not in k]
    assert not missing, missing
    return fn, sd, ds, model_cfg


def main(only=None):
    os.makedirs(OUT, exist_ok=True)
    for c in CASES:
        if only and c['case'] not in only:
            continue
        fn, sd, ds, model_cfg = build(c)
        rays = case_rays(c)
        tr = torch.from_numpy(rays)
        out = ref_shim.run_reference(fn, tr, fields=['render_weights'])
        payload = {
            'rays': rays,
            'rgb': out['rgb'].numpy().astype(np.float32),
            'recipe': np.frombuffer(json.dumps({
                'case': c['case'], 'model': c['model'], 'z_channels': c['z'], 'grid': c['grid'],
                'seed': c['seed'], 'density': c['density'], 'app_scale': c.get('app_scale', 1.0), 'dataset': ds,
                'mlp': c.get('mlp', 'default'), 'checksum': scenes.state_dict_checksum(sd)}).encode(), dtype=np.uint8),
        }
        if c.get('frame') is not None:
            # the frame's rays are a pure function of (model, frame): the fixture keeps their pixel indices only and
            # tests/helpers.py splices scenes.benchmark_rays(...)[frame_idx] back in at row frame_at
            at, idx = c['_frame_at'], c['_frame_idx']
            payload['rays'] = np.concatenate([rays[:at], rays[at + idx.shape[0]:]], 0)
            payload['frame_idx'] = idx
            payload['frame_at'] = np.asarray([at, c['frame'][0]], np.int32)
        if not c.get('rgb_only'):
            emb = ref_shim.run_reference_embed(fn, tr)
            Z = model_cfg.embedding.embeddings.ray_prediction_0.z_channels
            n = rays.shape[0]
            payload['render_weights'] = out['render_weights'].numpy().astype(np.float32)
            payload['points'] = emb['points'].numpy().reshape(n, Z, 3).astype(np.float32)
            payload['distances'] = emb['distances'].numpy().reshape(n, Z).astype(np.float32)
            payload['color_scale'] = emb['color_scale'].numpy().reshape(n, Z, 3).astype(np.float32)
            payload['color_shift'] = emb['color_shift'].numpy().reshape(n, Z, 3).astype(np.float32)
            if 'base_times' in emb:
                payload['base_times'] = emb['base_times'].numpy().reshape(n, Z)[:, 0].astype(np.float32)
        path = os.path.join(OUT, c['case'] + '.npz')
        np.savez_compressed(path, **payload)
        print(f"{c['case']}: {rays.shape[0]} rays, rgb mean {payload['rgb'].mean():.4f} std {payload['rgb'].std():.4f}, "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


def main_postfit(only=None):
    """The reference trained by the reference (make_fit_golden.fit: 200 Adam steps), then rendered by the reference on fresh rays."""
    import make_fit_golden as F
    for case, (fit_case, n_rays, ray_seed) in POSTFIT.items():
        if only and case not in only:
            continue
        model, grid, s_seed, t_seed, (H, W, frame), lr = F.CASES[fit_case]
        F.LR = lr
        cfg, ds = C.model_config(model), C.dataset_scalars(model)
        fit_rays = torch.from_numpy(np.ascontiguousarray(scenes.benchmark_rays(model, H, W, frame=frame), np.float32))
        teacher = F.build(model, grid, ds, scenes.make_state_dict(cfg, ds, grid, t_seed, 'dense', 1.0))
        target = ref_shim.run_reference(teacher, fit_rays)['rgb'].detach().clone()
        student_sd = scenes.make_state_dict(cfg, ds, grid, s_seed, 'dense', 1.0)
        fn = F.build(model, grid, ds, student_sd)
        fn.train()
        params = [p for n, p in fn.named_parameters() if 'dummy' not in n]
        opt = torch.optim.Adam(params, lr=lr)
        real_rand = torch.rand
        torch.rand = lambda *a, **k: torch.full((1,), 0.9)
        try:
            for step in range(F.N_STEPS):
                opt.zero_grad(set_to_none=True)
                with ref_shim.cpu_mode():
                    loss = ((fn(fit_rays)['rgb'] - target) ** 2).mean()
                loss.backward()
                opt.step()
        finally:
            torch.rand = real_rand
        fn.eval()
        trained = {k: v.detach().numpy().astype(np.float32) for k, v in fn.state_dict().items() if k in student_sd and not k.endswith('gridSize')}
        video = not model.startswith('donerf')
        full = scenes.benchmark_rays(model, 800, 800, frame=frame)
        idx = np.sort(np.random.default_rng(ray_seed).choice(full.shape[0], n_rays, replace=False))
        rays = np.ascontiguousarray(np.concatenate([full[idx], special_rays(video, 'z_plane' in model)], 0), np.float32)
        out = ref_shim.run_reference(fn, torch.from_numpy(rays))
        payload = {'rays': rays, 'rgb': out['rgb'].numpy().astype(np.float32),
                   'recipe': np.frombuffer(json.dumps({
                       'case': case, 'model': model, 'z_channels': None, 'grid': grid, 'seed': s_seed, 'density': 'dense', 'app_scale': 1.0,
                       'dataset': ds, 'mlp': 'default', 'postfit_of': fit_case, 'final_loss': float(loss.detach()),
                       'checksum': scenes.state_dict_checksum(student_sd)}).encode(), dtype=np.uint8)}
        for k, v in trained.items():
            payload['sd/' + k] = v
        path = os.path.join(OUT, case + '.npz')
        np.savez_compressed(path, **payload)
        moved = max(float(np.abs(trained[k] - student_sd[k]).max()) for k in trained if trained[k].size)
        print(f"{case}: {rays.shape[0]} rays, final loss {float(loss.detach()):.5f}, largest weight change {moved:.3f}, rgb mean {payload['rgb'].mean():.4f}, "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == '__main__':
    only = sys.argv[1:] or None
    main(only)
    main_postfit(only)
