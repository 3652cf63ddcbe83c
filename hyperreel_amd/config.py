"""Config surface: the reference's Hydra/OmegaConf `experiment.model` group as plain dicts.

The reference selects every hot-path component through `type:` strings in the
model YAML (conf/experiment/model/*.yaml) and reads a handful of dataset
scalars from `system.dm.train_dataset`.  This module
keeps that surface:

* `Cfg`              attr-dict with the access patterns the reference uses on
                     DictConfig (`'k' in cfg`, `cfg.k`, `cfg['k']`, assignment);
* `load_model_yaml`  reads any reference-style model YAML;
* `epoch_to_iter`    the `*_epoch(s)` -> `*_iter(s)` rewrite of
                     nlf/__init__.py:305-315 (utils/config_utils.py:32-38);
* `model_config`     built-in model groups with the same keys and values as the
                     shipped YAMLs of the BASELINE configs, composed from shared
                     pieces rather than stored as files;
* `n_to_reso`        utils/tensorf_utils.py:65-68 (final grid resolution).
"""
import copy
import re

import numpy as np
import yaml


class Cfg(dict):
    """dict with attribute access (stands in for OmegaConf's DictConfig)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_cfg(o):
    if isinstance(o, dict):
        return Cfg({k: to_cfg(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return [to_cfg(v) for v in o]
    return o


def to_plain(o):
    if isinstance(o, dict):
        return {k: to_plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [to_plain(v) for v in o]
    return o


class _Loader(yaml.SafeLoader):
    """SafeLoader that also reads `1e-3` as a float, as OmegaConf does (YAML 1.1
    requires a dot in the mantissa, so PyYAML alone yields the string '1e-3')."""


_Loader.add_implicit_resolver(
    'tag:yaml.org,2002:float',
    re.compile(r'^[-+]?(?:[0-9][0-9_]*)(?:\.[0-9_]*)?[eE][-+]?[0-9]+$'),
    list('-+0123456789'))


def load_model_yaml(path):
    with open(path) as f:
        return to_cfg(yaml.load(f, Loader=_Loader))


_EPOCH_KEYS = ['max_freq', 'wait', 'stop', 'falloff', 'window', 'no_bias',
               'window_bias', 'window_bias_start', 'decay', 'warmup']


def epoch_to_iter(cfg, iters_per_epoch):
    """Adds `<k>_iter(s)` next to every `<k>_epoch(s)` key, in place.

    Follows nlf/__init__.py:305-315: a key that matches is rewritten and not
    descended into; list values are lists of [start, end] pairs."""
    finds = [f'{k}_epoch' for k in _EPOCH_KEYS] + [f'{k}_epochs' for k in _EPOCH_KEYS]

    def walk(c, find):
        if not isinstance(c, dict):
            return
        for key in list(c.keys()):
            if key == find:
                v = c[key]
                nk = key.replace('epoch', 'iter')
                if isinstance(v, list):
                    c[nk] = [[x * iters_per_epoch for x in li] for li in v]
                else:
                    c[nk] = v * iters_per_epoch
            else:
                walk(c[key], find)

    for f in finds:
        walk(cfg, f)
    return cfg


def n_to_reso(n_voxels, aabb):
    """utils/tensorf_utils.py:65-68 evaluated the way torch does (float32)."""
    lo = np.asarray(aabb[0], np.float32)
    hi = np.asarray(aabb[1], np.float32)
    size = hi - lo
    voxel = np.float32(np.power(np.float32(np.prod(size) / np.float32(n_voxels)), np.float32(1.0 / 3.0)))
    return [int(v) for v in (size / voxel).astype(np.int64)]


# --------------------------------------------------------------------------- built-in groups
def _sig(shift, wait_epochs):
    return {'type': 'ease_value', 'start_value': 1.0, 'window_epochs': 3, 'wait_epochs': wait_epochs,
            'activation': {'type': 'sigmoid', 'shift': shift}}


def _affine_head():
    return {'type': 'ease_value', 'start_value': 0.0, 'window_epochs': 0, 'wait_epochs': 0,
            'activation': {'type': 'identity', 'shift': 0.0, 'inner_fac': 1.0, 'outer_fac': 1.0}}


def _windowed(n_freqs, with_mult=True):
    pe = {'type': 'windowed'}
    if with_mult:
        pe['freq_multiplier'] = 2.0
    pe.update({'n_freqs': n_freqs, 'wait_iters': 0, 'max_freq_epoch': 0, 'exclude_identity': False})
    return pe


def _flow_stage():
    fac = {'type': 'identity', 'fac': 0.25}
    return {'type': 'advect_points', 'use_spatial_flow': True, 'use_angular_flow': False,
            'out_flow_field': 'raw_flow', 'flow_scale': 0.0,
            'spatial_flow_activation': dict(fac),
            'angular_flow_rotation_activation': dict(fac),
            'angular_flow_anchor_activation': dict(fac)}


_VIDEO_FIELDS = ['points', 'distances', 'base_times', 'time_offset', 'times', 'viewdirs', 'weights',
                 'color_transform_global', 'color_scale_global', 'color_shift_global',
                 'color_transform', 'color_scale', 'color_shift']


def _model(*, video, ray_param, ray_pe_freqs, time_pe_mult, z, z_val_channels, flow_fac, sigma_shift,
           offset_fac, intersect, aabb, n_init, n_final, alpha_list, n_lamb, shading):
    params = {'ray': {'start': 0, 'end': 6, 'param': ray_param, 'pe': _windowed(ray_pe_freqs)}}
    if video:
        params['time'] = {'start': 7, 'end': 8, 'param': {'n_dims': 1, 'fn': 'identity'},
                          'pe': _windowed(2, with_mult=time_pe_mult)}
    outputs = {'z_vals': {'channels': z_val_channels}}
    if video:
        outputs['spatial_flow'] = {'channels': 3, 'activation': {'type': 'identity', 'outer_fac': flow_fac}}
    outputs['sigma'] = {'channels': 1, 'activation': _sig(sigma_shift, 0)}
    outputs['point_sigma'] = {'channels': 1, 'activation': _sig(4.0, 1)}
    outputs['point_offset'] = {'channels': 3, 'activation': {'type': 'tanh', 'outer_fac': offset_fac}}
    outputs['color_scale'] = {'channels': 3, 'activation': _affine_head()}
    outputs['color_shift'] = {'channels': 3, 'activation': _affine_head()}

    emb = {
        'ray_prediction_0': {
            'type': 'ray_prediction', 'params': params,
            'net': {'type': 'base', 'group': 'embedding_impl', 'depth': 6, 'hidden_channels': 256, 'skips': [3]},
            'z_channels': z, 'outputs': outputs},
        'ray_intersect_0': {'type': 'ray_intersect', 'z_channels': z, 'intersect': intersect},
    }
    if video:
        emb['flow_0'] = _flow_stage()
        emb['point_offset_0'] = {'type': 'point_offset', 'in_density_field': 'point_sigma', 'use_sigma': True}
        emb['add_point_outputs_0'] = {'type': 'add_point_outputs', 'extra_outputs': ['viewdirs', 'times']}
        emb['extract_fields'] = {'type': 'extract_fields', 'fields': list(_VIDEO_FIELDS)}
    else:
        emb['point_offset_0'] = {'type': 'point_offset', 'use_sigma': True}
        emb['add_point_outputs_0'] = {'type': 'add_point_outputs', 'extra_outputs': ['viewdirs']}
        emb['extract_fields'] = {'type': 'extract_fields',
                                 'fields': ['points', 'distances', 'viewdirs', 'weights', 'color_scale', 'color_shift']}

    net = {'type': 'tensor_vm_split_time' if video else 'tensor_vm_split_no_sample',
           'white_bg': 0, 'black_bg': 0, 'fea2denseAct': 'relu', 'distance_scale': 16.0, 'density_shift': 0.0,
           'aabb': aabb, 'N_voxel_init': n_init, 'N_voxel_final': n_final,
           'upsamp_list': [4000, 6000, 8000, 10000, 12000], 'lr_upsample_reset': True,
           'update_AlphaMask_list': alpha_list, 'rm_weight_mask_thre': 0, 'alpha_mask_thre': 1e-3,
           'n_lamb_sigma': list(n_lamb), 'n_lamb_sh': list(n_lamb),
           'shadingMode': shading, 'data_dim_color': 3 if shading == 'RGB' else 27}
    if video:
        net['densityMode'] = 'Density'
    return {'type': 'lightfield', 'render': {'type': 'lightfield'},
            'param': {'n_dims': 6, 'fn': 'identity'},
            'embedding': {'type': 'ray_point', 'embeddings': emb},
            'color': {'type': 'base', 'net': net}}


_PLUECKER = {'n_dims': 6, 'fn': 'pluecker', 'direction_multiplier': 1.0, 'moment_multiplier': 1.0}
_HALF = {'type': 'identity', 'fac': 0.5}


def _primitive(kind, outward):
    return {'type': kind, 'sort': True, 'outward_facing': outward, 'use_disparity': False, 'max_axis': False,
            'use_sigma': True, 'out_points': 'raw_points', 'out_distance': 'raw_distance',
            'use_dataset_bounds': True, 'origin_scale_factor': 0.0,
            'contract': {'type': 'mipnerf', 'contract_samples': True, 'use_dataset_bounds': True},
            'activation': dict(_HALF)}


def _z_plane(contract=None):
    c = {'type': 'z_plane', 'sort': True, 'outward_facing': False, 'use_disparity': False, 'use_sigma': True,
         'out_points': 'raw_points', 'out_distance': 'raw_distance', 'initial': -1.0, 'end': 1.0}
    if contract is not None:
        c['contract'] = contract
    c['activation'] = dict(_HALF)
    return c


def _donerf(kind):
    return _model(video=False, ray_param=dict(_PLUECKER), ray_pe_freqs=1, time_pe_mult=False, z=32,
                  z_val_channels=4, flow_fac=None, sigma_shift=4.0, offset_fac=0.125,
                  intersect=_primitive(kind, False), aabb=[[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]],
                  n_init=3375000, n_final=216000000, alpha_list=[4000, 8000], n_lamb=[8, 4, 4], shading='RGB')


_BUILDERS = {
    'donerf_sphere': lambda: _donerf('sphere'),
    'donerf_cylinder': lambda: _donerf('cylinder'),
    'technicolor_z_plane': lambda: _model(
        video=True, ray_param={'n_dims': 4, 'fn': 'two_plane'}, ray_pe_freqs=0, time_pe_mult=False, z=32,
        z_val_channels=1, flow_fac=0.25, sigma_shift=4.0, offset_fac=0.25, intersect=_z_plane(),
        aabb=[[-2.0, -2.0, -1.0], [2.0, 2.0, 1.0]], n_init=2097152, n_final=512000000,
        alpha_list=[4000, 8000], n_lamb=[8, 0, 0], shading='SH'),
    'neural_3d_z_plane': lambda: _model(
        video=True, ray_param=dict(_PLUECKER), ray_pe_freqs=1, time_pe_mult=True, z=64,
        z_val_channels=1, flow_fac=4.0, sigma_shift=1.0, offset_fac=0.25,
        intersect=_z_plane({'type': 'mipnerf', 'contract_samples': True,
                            'contract_start_radius': 1.0, 'contract_end_radius': 8.0}),
        aabb=[[-2.0, -1.5, -1.25], [2.0, 1.5, 1.25]], n_init=2097152, n_final=262144000,
        alpha_list=[], n_lamb=[8, 4, 4], shading='SH'),
    'immersive_sphere': lambda: _model(
        video=True, ray_param=dict(_PLUECKER), ray_pe_freqs=1, time_pe_mult=False, z=32,
        z_val_channels=4, flow_fac=1.0, sigma_shift=4.0, offset_fac=0.25,
        intersect=_primitive('sphere', True), aabb=[[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]],
        n_init=2097152, n_final=262144000, alpha_list=[4000, 8000], n_lamb=[8, 4, 4], shading='SH'),
}

MODEL_NAMES = tuple(_BUILDERS)

# Dataset scalars the hot path reads from `system.dm.train_dataset`.
# DoNeRF's come from the scene's dataset_info.json, which is not available: the
# synthetic DoNeRF scene fixes them to the survey's values.
DATASET_SCALARS = {
    'donerf': {'near': 0.5, 'far': 20.0, 'depth_range': [0.5, 20.0], 'num_keyframes': 1, 'num_frames': 1},
    'technicolor': {'near': 0.5, 'far': 20.0, 'depth_range': [0.5, 20.0], 'num_keyframes': 12, 'num_frames': 50},
    'neural_3d': {'near': 0.5, 'far': 20.0, 'depth_range': [0.5, 20.0], 'num_keyframes': 12, 'num_frames': 50},
    'immersive': {'near': 1.0, 'far': 10.0, 'depth_range': [2.0, 10.0], 'num_keyframes': 12, 'num_frames': 50},
}


def model_config(name, z_channels=None):
    """Built-in `experiment.model` group (same keys/values as the shipped YAML).

    z_channels overrides `z_channels` of both ray_prediction_0 and
    ray_intersect_0 (BASELINE config 1 uses 16)."""
    cfg = to_cfg(_BUILDERS[name]())
    if z_channels is not None:
        cfg.embedding.embeddings.ray_prediction_0.z_channels = int(z_channels)
        cfg.embedding.embeddings.ray_intersect_0.z_channels = int(z_channels)
    return cfg


def dataset_scalars(name):
    for k, v in DATASET_SCALARS.items():
        if name.startswith(k):
            return copy.deepcopy(v)
    raise KeyError(name)


def final_grid_size(cfg):
    n = cfg['color']['net']
    return n_to_reso(n['N_voxel_final'], n['aabb'])
