"""Coverage sweep: one small golden per SHIPPED model YAML that the HIP backend accepts.

    python oracle/refgen/make_sweep.py [name ...]

TEST INFRASTRUCTURE ONLY (authoring container; imports the reference through ref_shim.py).
For every conf/experiment/model/*.yaml that `hyperreel_amd.plan.compile_config` accepts, the
reference itself is built from that YAML (grid overridden to a small one), loaded with seeded
weights and run on seeded rays.  The fixture tests/golden/sweep/<name>.npz stores the rays, the
reference's rgb and the recipe -- including the parsed `experiment.model` group, because the GPU
box has no /root/reference to read the YAML from.  YAMLs the backend rejects are listed with the
reason in tests/golden/sweep/coverage.json (read by tests and DESIGN.md).
"""
import glob
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
from hyperreel_amd import plan, scenes  # noqa: E402
from make_golden import special_rays  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'sweep')
GRID = [28, 24, 20]

# dataset scalars the reference constructors read (synthetic)
DEFAULT_DS = {'near': 0.5, 'far': 20.0, 'depth_range': [0.5, 20.0], 'num_keyframes': 12, 'num_frames': 50}
# voxel_grid with use_dataset_bounds reads the dataset's bounding box (voxel.py:27-29)
BBOX = {'bbox_min': [-1.5, -1.25, -1.75], 'bbox_max': [1.5, 1.75, 1.25]}


def dataset_for(name):
    try:
        ds = C.dataset_scalars(name)
    except KeyError:
        ds = dict(DEFAULT_DS)
    if 'voxel' in name:
        ds.update(BBOX)
    if name == 'immersive_z_plane':             # ColorTransformEmbedding reads these (point.py:576-577)
        ds.update({'total_images_per_frame': 5, 'val_all': True})
    return ds


def sweep_rays(cfg, seed):
    video = cfg.color.net.type == 'tensor_vm_split_time' or any(
        e.get('type') == 'advect_points' for e in cfg.embedding.embeddings.values())
    isect = [e for e in cfg.embedding.embeddings.values() if e.get('type') == 'ray_intersect'][0]
    z_plane = isect.intersect.type == 'z_plane'
    if z_plane:
        r = scenes.random_rays(88, seed, video, pos_mean=(0, 0, 1.0), pos_std=0.15, dir_mean=(0, 0, -1.2), dir_std=0.5)
    elif isect.intersect.type in ('sphere_new', 'cylinder_new'):
        # half of the rays start well off-centre so that they MISS the inner primitives and take the
        # sample-recycling branch (primitive.py:536-541)
        r = np.concatenate([scenes.random_rays(44, seed, video), scenes.random_rays(44, seed + 1000, video, pos_std=1.2)], 0)
    else:
        r = scenes.random_rays(88, seed, video)
    ray_dim = r.shape[1]
    sp = special_rays(ray_dim == 8, z_plane)
    r = np.ascontiguousarray(np.concatenate([r, sp], 0), np.float32)
    if any(e.get('type') == 'color_transform' for e in cfg.embedding.embeddings.values()):
        if r.shape[1] == 6:                     # [o, d, camera id, t]
            r = np.concatenate([r, np.zeros((r.shape[0], 2), np.float32)], -1)
        r[:, 6] = (np.arange(r.shape[0]) % 5).astype(np.float32) + np.float32(0.2)   # round() -> 0..4
    return np.ascontiguousarray(r, np.float32)


# Options the reference implements but no shipped YAML selects: a shipped YAML plus an edit, run through the
# reference like the others.  (name, base YAML, edit applied to both the reference's and our config)
def _isect(cfg):
    return [e for e in cfg.embedding.embeddings.values() if e.get('type') == 'ray_intersect'][0].intersect


def _pred(cfg):
    return [e for e in cfg.embedding.embeddings.values() if e.get('type') == 'ray_prediction'][0]


def _v_cylinder_new(cfg):
    ic = _isect(cfg)
    ic.type = 'cylinder_new'
    ic.origin_scale_factor = 0.3
    ic.resize_scale_factor = 0.25
    _pred(cfg).outputs.z_vals.channels = 8


def _v_sphere_new_origins_only(cfg):          # origins live, resize dead: head-column compaction with a gap
    ic = _isect(cfg)
    ic.origin_scale_factor = 0.4
    ic.resize_scale_factor = 0.0


def _v_z_depth(cfg):
    ic = _isect(cfg)
    ic.contract = {'type': 'z_depth', 'contract_samples': True, 'contract_end_radius': 6.0}


def _v_donerf_contract(cfg):                  # DoNeRFContract (contract.py:195-240) with the dataset's bounds: power = log2(r1 / r0), a general powf
    _isect(cfg).contract = {'type': 'donerf', 'contract_samples': True, 'use_dataset_bounds': True}


def _v_donerf_contract_pow2(cfg):             # ... and with its defaults: power 2 (torch.pow takes x * x / sqrt), fac = 4 / end radius
    _isect(cfg).contract = {'type': 'donerf', 'contract_samples': True, 'contract_end_radius': 40.0}


def _v_donerf_contract_pow3(cfg):             # ... power 3: ATen's pow kernel takes x * x * x for the inverse contraction, powf(., 1/3) for the points
    _isect(cfg).contract = {'type': 'donerf', 'contract_samples': True, 'contract_end_radius': 40.0, 'power': 3.0}


def _v_color_transform_global_head(cfg):      # transform_color_one fed from the MLP head instead of the per-camera table (tensorf_no_sample.py:242-243)
    out = _pred(cfg).outputs
    spec = out.pop('color_scale_global')
    out['color_transform_global'] = C.to_cfg({**C.to_plain(spec), 'channels': 9}) if hasattr(C, 'to_cfg') else spec
    out['color_transform_global']['channels'] = 9
    for e in cfg.embedding.embeddings.values():
        if e.get('type') == 'extract_fields':
            e.fields = [('color_transform_global' if f == 'color_scale_global' else f) for f in e.fields]


def _v_voxel_outward(cfg):
    ic = _isect(cfg)
    ic.outward_facing = True
    ic.use_dataset_bounds = False
    ic.initial = [0.1, 0.15, 0.2]
    ic.end = [1.9, 1.7, 1.8]
    ic.z_scale = [0.05, 0.04, 0.06]


def _v_mask_off_unsorted(cfg):
    ic = _isect(cfg)
    ic.mask = {'stop_iters': 5}
    ic.sort = False


def _v_deformable_3axes(cfg):                 # default start normals (x, y, z), 16 planes per axis
    st = [e for e in cfg.embedding.embeddings.values() if e.get('type') == 'ray_intersect'][0]
    st.z_channels = 48
    _pred(cfg).z_channels = 48
    ic = st.intersect
    ic.pop('start_normal')
    ic.normal_scale_factor = 0.3
    ic.initial = [-1.0, -0.8, -0.2]
    ic.end = [1.0, 0.9, 1.6]


def _v_pe_window(cfg):
    """a WindowedPE schedule on the ray encoding (no runnable shipped YAML has one; shiny_z_depth's sits in an MLP the
    reference cannot build): 2 epochs = 8000 iterations over the encoding's frequencies"""
    for p in cfg.embedding.embeddings.ray_prediction_0.params.values():
        if 'pe' in p:
            p.pe.max_freq_epoch = 2


# (name, base YAML, edit, [training iteration the schedules are evaluated at])
VARIANTS = [
    ('variant_deformable_3axes', 'shiny_z_deformable', _v_deformable_3axes),
    ('variant_cylinder_new', 'bom_cylinder', _v_cylinder_new),
    ('variant_sphere_new_origins_only', 'immersive_sphere_new', _v_sphere_new_origins_only),
    ('variant_z_depth_contract', 'llff_z_plane', _v_z_depth),
    ('variant_voxel_outward', 'donerf_voxel', _v_voxel_outward),
    ('variant_color_transform_global_head', 'catacaustics_distance', _v_color_transform_global_head),
    ('variant_donerf_contract', 'donerf_sphere', _v_donerf_contract),
    ('variant_donerf_contract_pow2', 'donerf_cylinder', _v_donerf_contract_pow2),
    ('variant_donerf_contract_pow3', 'donerf_sphere', _v_donerf_contract_pow3),
    ('variant_mask_off_unsorted', 'donerf_sphere', _v_mask_off_unsorted),
    # inside the activation / encoding warm-up windows (EaseValue, activations.py:462-496; WindowedPE, pe.py:166-208)
    ('variant_ease_iter2000', 'donerf_sphere', None, 2000),
    ('variant_ease_iter6000', 'immersive_sphere', None, 6000),
    ('variant_ease_iter0', 'technicolor_z_plane', None, 0),
    ('variant_pe_window_iter3000', 'donerf_sphere', _v_pe_window, 3000),
    ('variant_mask_stop_iter3', 'donerf_sphere', _v_mask_off_unsorted, 3),       # before mask.stop_iters = 5: the mask is still on
]


def main(only=None, force=False):
    """force: also run YAMLs the plan compiler rejects and only report oracle-vs-reference error
    (used while widening the oracle ahead of the kernels; writes nothing for those)."""
    os.makedirs(OUT, exist_ok=True)
    coverage = {}
    names = sorted(os.path.basename(p)[:-5] for p in glob.glob(f'{ref_shim.REF}/conf/experiment/model/*.yaml'))
    jobs = [(n, n, None, None) for n in names] + [(v + (None,))[:4] for v in VARIANTS]
    for i, (name, base, edit, iteration) in enumerate(jobs):
        if only and name not in only:
            continue
        path = f'{ref_shim.REF}/conf/experiment/model/{base}.yaml'
        ds = dataset_for(base)
        raw = C.load_model_yaml(path)
        if raw is None:
            coverage[name] = {'status': 'rejected', 'reason': 'the shipped YAML is empty'}
            continue
        raw.color.net.grid_size = C.to_cfg({'start': list(GRID), 'end': list(GRID)})
        if edit:
            edit(raw)
            raw = C.to_cfg(C.to_plain(raw))
        model_cfg = C.epoch_to_iter(C.to_cfg(C.to_plain(raw)), 4000)
        rejected = None
        try:
            plan.compile_model(model_cfg, ds, GRID, iteration=iteration)
        except (NotImplementedError, ValueError) as e:
            coverage[name] = {'status': 'rejected', 'reason': str(e)}
            print(f'{name:36s} rejected: {e}')
            rejected = e
            if not force:
                # record whether the reference itself can build and run this YAML at all
                try:
                    def ov(cfg):
                        cfg.color.net.grid_size = ref_shim.to_attr({'start': list(GRID), 'end': list(GRID)})
                    fn = ref_shim.build_reference(ref_shim.load_model_cfg(base, ov), ds)
                    v = any(e.get('type') == 'advect_points' for e in raw.embedding.embeddings.values()) or raw.color.net.type == 'tensor_vm_split_time'
                    ref_shim.run_reference(fn, torch.from_numpy(scenes.random_rays(8, 1, v)))
                    coverage[name]['reference_runs'] = True
                except Exception as e2:                      # noqa: BLE001 -- any failure of the reference counts
                    coverage[name]['reference_runs'] = False
                    coverage[name]['reference_error'] = f'{type(e2).__name__}: {str(e2).splitlines()[0][:160]}'
                continue

        def overrides(cfg):
            cfg.color.net.grid_size = ref_shim.to_attr({'start': list(GRID), 'end': list(GRID)})
            if edit:
                edit(cfg)
                for k, v in list(_isect(cfg).items()):      # plain dicts written by an edit -> attr dicts
                    _isect(cfg)[k] = ref_shim.to_attr(v)
        ref_cfg = ref_shim.load_model_cfg(base, overrides)
        fn = ref_shim.build_reference(ref_cfg, ds) if iteration is None else ref_shim.build_reference(ref_cfg, ds, iteration=iteration)
        seed = 100 + i
        sd = scenes.make_state_dict(model_cfg, ds, GRID, seed, 'dense', 1.0)
        own = dict(fn.state_dict())
        with torch.no_grad():
            for k, v in sd.items():
                if k.endswith('gridSize'):
                    assert own[k].tolist() == v.tolist(), (k, own[k], v)
                    continue
                assert tuple(own[k].shape) == tuple(v.shape), (name, k, own[k].shape, v.shape)
                own[k].copy_(torch.from_numpy(v))
        missing = [k for k in own if k not in sd and 'dummy_layer' not in k]
        assert not missing, (name, missing)
        rays = sweep_rays(model_cfg, seed)
        try:
            out = ref_shim.run_reference(fn, torch.from_numpy(rays))
        except Exception as e:          # noqa: BLE001 -- a shipped YAML the reference itself cannot run
            coverage[name] = {'status': 'reference_fails', 'reason': str(e).splitlines()[0]}
            print(f'{name:36s} the reference raises: {coverage[name]["reason"]}')
            continue
        rgb = out['rgb'].numpy().astype(np.float32)
        dropped = 0
        if np.isnan(rgb).any():         # DoNeRFContract.contract_points is 0 / 0 for a ray that starts at the centre: the reference returns NaN
            keep = ~np.isnan(rgb).any(axis=1)
            dropped = int((~keep).sum())
            rays, rgb = np.ascontiguousarray(rays[keep]), np.ascontiguousarray(rgb[keep])
        if force:
            sys.path.insert(0, os.path.join(ROOT, 'oracle'))
            from hyperreel_oracle import HyperReelOracle
            try:
                got = HyperReelOracle(model_cfg, ds, sd, iteration=iteration).render(rays)['rgb']
                print(f'{name:36s} oracle vs reference: L-inf {np.abs(got - rgb).max():.3e} (rgb std {rgb.std():.3f})')
            except NotImplementedError as e:
                print(f'{name:36s} oracle: NotImplementedError {e}')
        if rejected is not None:
            continue
        recipe = {'case': 'sweep/' + name, 'model': name, 'model_cfg': C.to_plain(raw), 'z_channels': None,
                  'grid': GRID, 'seed': seed, 'density': 'dense', 'app_scale': 1.0, 'dataset': ds,
                  'checksum': scenes.state_dict_checksum(sd)}
        if iteration is not None:
            recipe['iter'] = iteration
        if dropped:
            recipe['dropped_nan_rays'] = dropped
        np.savez_compressed(os.path.join(OUT, name + '.npz'), rays=rays, rgb=rgb,
                            recipe=np.frombuffer(json.dumps(recipe).encode(), dtype=np.uint8))
        coverage[name] = {'status': 'golden', 'rgb_std': float(rgb.std())}
        print(f'{name:36s} golden: {rays.shape[0]} rays, rgb mean {rgb.mean():.4f} std {rgb.std():.4f}')
    cov_path = os.path.join(OUT, 'coverage.json')
    if only and not force and os.path.exists(cov_path):      # a partial run updates its own entries only
        with open(cov_path) as f:
            coverage = {**json.load(f), **coverage}
    if not force:
        with open(cov_path, 'w') as f:
            json.dump(coverage, f, indent=1, sort_keys=True)
    n_ok = sum(1 for k, v in coverage.items() if v['status'] == 'golden' and not k.startswith('variant_'))
    n_all = sum(1 for k in coverage if not k.startswith('variant_'))
    print(f'{n_ok} of {n_all} shipped model YAMLs covered (+ {sum(1 for k in coverage if k.startswith("variant_"))} variants)')


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != '--force']
    main(args or None, force='--force' in sys.argv[1:])
