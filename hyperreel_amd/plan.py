"""Compiles the reference's model YAML (+ five dataset scalars) into the flat `hr_config`
that libhyperreel_hip.so consumes (include/hyperreel_hip.h).

This is the host-side half of the drop-in boundary: everything the reference's module
constructors derive at build time is derived here, in the reference's own precision and
order, so the kernels only see numbers:

  RayPredictionEmbedding.__init__   nlf/embedding/ray.py:213-315   (param groups, PE, head layout, depth-2)
  Intersect.__init__                nlf/intersect/base.py:52-126   (near/far, sort, contract, activation)
  IntersectZPlane.__init__          nlf/intersect/z.py:16-75       (anchor samples, z_scale)
  IntersectSphereOld/CylinderOld    nlf/intersect/primitive.py:181-233, 366-418
  MIPNeRFContract.__init__          nlf/contract.py:113-141
  AdvectPointsEmbedding.__init__    nlf/embedding/point.py:741-778
  PointOffsetEmbedding.__init__     nlf/embedding/point.py:338-369
  TensorBase.__init__               nlf/nets/tensorf_base.py:137-248
  TensorVMKeyframeTime.__init__     nlf/nets/tensorf_dynamic.py:45-77

Anything outside the hot-path scope raises NotImplementedError with
the offending key -- never a silent fallback.
"""
import ctypes as C

import numpy as np

F32 = np.float32

HR_MAX_Z = 256          # size of hr_config.samples
HR_KERNEL_MAX_Z = 256   # what the sample kernel handles (a 256-thread block per ray at most)
HR_MAX_GROUPS = 4
HR_MAX_LAYERS = 8
HR_MAX_MLP_IN = 64

ACT = {'identity': 0, 'sigmoid': 1, 'tanh': 2}
PARAM = {'identity': 0, 'pluecker': 1, 'two_plane': 2}
PE = {None: 0, 'windowed': 1, 'basic': 2}
ISECT = {'z_plane': 0, 'sphere': 1, 'cylinder': 2, 'sphere_new': 3, 'cylinder_new': 4, 'euclidean_distance_unified': 5,
         'voxel_grid': 6, 'deformable_voxel_grid': 7}
ISECT_Z_CHANNELS = {0: 1, 1: 4, 2: 4, 3: 8, 4: 8, 5: 1, 6: 1, 7: 4}     # z_vals channels each type reads per sample
CONTRACT = {'identity': 0, 'mipnerf': 1, 'bbox': 2, 'z_depth': 2, 'donerf': 3}   # bbox and z_depth share the affine kernel path
DENSITY = {'relu': 0, 'softplus': 1, 'relu_abs': 2}
SHADING = {'RGB': 0, 'SH': 1}


class hr_act(C.Structure):
    _fields_ = [('type', C.c_int32), ('inner', C.c_float), ('shift', C.c_float), ('outer', C.c_float), ('add', C.c_float)]


class hr_param_group(C.Structure):
    _fields_ = [('start', C.c_int32), ('end', C.c_int32), ('fn', C.c_int32), ('origin', C.c_float * 3),
                ('a', C.c_float), ('b', C.c_float), ('pe_type', C.c_int32), ('pe_n_freqs', C.c_int32),
                ('pe_exclude_identity', C.c_int32), ('pe_freq_mult', C.c_float), ('pe_base_mult', C.c_float),
                ('pe_weight', C.c_float * 8)]


class hr_head_field(C.Structure):
    _fields_ = [('offset', C.c_int32), ('channels', C.c_int32), ('act', hr_act)]


class hr_config(C.Structure):
    _fields_ = [
        ('ray_dim', C.c_int32), ('n_groups', C.c_int32), ('groups', hr_param_group * HR_MAX_GROUPS),
        ('mlp_in', C.c_int32), ('mlp_layers', C.c_int32), ('mlp_hidden', C.c_int32), ('mlp_skip_mask', C.c_int32),
        ('leaky_slope', C.c_float), ('z_channels', C.c_int32), ('preds_per_z', C.c_int32),
        ('f_z_vals', hr_head_field), ('f_isect_sigma', hr_head_field), ('f_offset_sigma', hr_head_field),
        ('f_point_offset', hr_head_field), ('f_color_scale', hr_head_field), ('f_color_shift', hr_head_field),
        ('f_spatial_flow', hr_head_field), ('f_color_scale_global', hr_head_field), ('f_color_shift_global', hr_head_field),
        ('isect_type', C.c_int32), ('isect_origin', C.c_float * 3), ('near', C.c_float), ('far', C.c_float),
        ('z_act', hr_act), ('sort', C.c_int32), ('samples', C.c_float * HR_MAX_Z), ('z_scale', C.c_float),
        ('origin_scale', C.c_float), ('origin_initial', C.c_float * 3),
        ('resize_scale', C.c_float), ('resize_initial', C.c_float * 3),
        ('voxel_scale', C.c_float * 3), ('isect_outward', C.c_int32),
        ('dvg_axes', C.c_int32), ('dvg_normals', C.c_float * 9), ('dvg_normal_scale', C.c_float), ('isect_mask_off', C.c_int32),
        ('contract_type', C.c_int32), ('contract_samples', C.c_int32),
        ('c_r0', C.c_float), ('c_r_inv_end', C.c_float), ('c_r_scale', C.c_float),
        ('c_d0', C.c_float), ('c_d_inv_end', C.c_float), ('c_d_scale', C.c_float),
        ('c_aff_min', C.c_float * 3), ('c_aff_size', C.c_float * 3), ('c_aff_fac', C.c_float),
        ('c_pow_fac', C.c_float), ('c_pow_power', C.c_float), ('c_pow_inv_power', C.c_float),
        ('advect', C.c_int32), ('use_spatial_flow', C.c_int32), ('flow_fac', C.c_float), ('flow_inv_fac', C.c_float),
        ('flow_kmax', C.c_float), ('flow_act', hr_act),
        ('point_offset', C.c_int32), ('offset_act', hr_act),
        ('video', C.c_int32), ('aabb', C.c_float * 6), ('inv_size', C.c_float * 3), ('grid', C.c_int32 * 3),
        ('num_keyframes', C.c_int32), ('n_den', C.c_int32 * 3), ('n_app', C.c_int32 * 3), ('app_dim', C.c_int32),
        ('shading', C.c_int32), ('distance_scale', C.c_float), ('weight_thresh', C.c_float),
        ('density_act', C.c_int32), ('density_shift', C.c_float), ('time_scale', C.c_float), ('time_offset', C.c_float),
        ('white_bg', C.c_int32), ('mlp_precision', C.c_int32), ('grid_dtype', C.c_int32),
        ('color_table_views', C.c_int32), ('color_table_t_act', hr_act), ('color_table_s_act', hr_act),
        ('casc_in_z', C.c_int32), ('casc_row_dim', C.c_int32), ('casc_n_inputs', C.c_int32),
        ('casc_input_kind', C.c_int32 * 4), ('casc_input_dim', C.c_int32 * 4),
    ]


class hr_camera(C.Structure):
    _fields_ = [('c2w', C.c_float * 12), ('fx', C.c_float), ('fy', C.c_float), ('cx', C.c_float), ('cy', C.c_float),
                ('width', C.c_int32), ('height', C.c_int32), ('cam_id', C.c_float), ('time', C.c_float)]


class hr_fields(C.Structure):
    _fields_ = [('distances_dev', C.c_void_p), ('points_dev', C.c_void_p), ('sigma_dev', C.c_void_p),
                ('weights_dev', C.c_void_p), ('head_dev', C.c_void_p)]


# --------------------------------------------------------------------------- small helpers
# Training iteration the schedules are evaluated at while compile_config runs (None: converged, every weight 1)
_ITERATION = None
# Bounding box of the feature grids when it is not the YAML's: TensorBase keeps `aabb` as a buffer that `shrink`
# replaces during training (nlf/nets/tensorf_base.py:1191-1232) and that checkpoints therefore carry
_AABB = None


def ease_weight(cfg, iteration):
    """EaseValue.weight at `iteration` (nlf/activations.py:462-496): cur = iteration - wait_iters; 1 once
    cur >= window_iters, 0 while waiting on a zero-length window, else clamp(cur / window_iters, 0, 1)."""
    if iteration is None:
        return 1.0
    cur = iteration - cfg.get('wait_iters', 0.0)
    window = cfg.get('window_iters', 0.0)
    if cur >= window:
        return 1.0
    if window == 0:
        return 0.0
    return min(max(float(cur) / window, 0.0), 1.0)


def windowed_pe_weights(pe, iteration):
    """WindowedPE.weight(j + window_identity) for j < n_freqs (nlf/pe.py:166-208)."""
    n = int(pe['n_freqs'])
    max_freq_iter = float(pe.get('max_freq_iter', 0))
    wait = float(pe.get('wait_iters', 0))
    window_identity = 1 if pe.get('window_identity', False) else 0
    if iteration is None or n == 0 or max_freq_iter == 0:
        return [1.0] * n
    window_after = max_freq_iter / n
    if window_identity:
        windows = [(wait, window_after + wait)] + [(window_after * i + wait, window_after * (i + 1) + wait) for i in range(1, n + 1)]
        max_freq_iter = (n + 1) * window_after
    else:
        windows = [(window_after * i + wait, window_after * (i + 1) + wait) for i in range(n)]
    out = []
    for j in range(n):
        jj = j + window_identity
        cur = iteration - wait
        if cur < 0:
            out.append(0.0)
        elif iteration > max_freq_iter:
            out.append(1.0)
        elif windows[jj][1] - windows[jj][0] == 0:
            out.append(1.0 if iteration >= windows[jj][0] else 0.0)
        else:
            alpha = (cur - windows[jj][0]) / float(windows[jj][1] - windows[jj][0])
            out.append(float((1.0 - np.cos(np.pi * np.clip(alpha, 0.0, 1.0))) / 2))
    return out


class at_iteration:
    """with at_iteration(i): compile_* evaluates the EaseValue / WindowedPE schedules at training iteration i
    (None: converged).  INRSystem.set_train_iter (nlf/__init__.py:608-614) is the reference's equivalent."""

    def __init__(self, iteration, aabb='inherit'):
        self.iteration, self.aabb = iteration, aabb

    def __enter__(self):
        global _ITERATION, _AABB
        self.prev = (_ITERATION, _AABB)
        _ITERATION = self.iteration
        if not isinstance(self.aabb, str):
            _AABB = self.aabb
        return self

    def __exit__(self, *exc):
        global _ITERATION, _AABB
        _ITERATION, _AABB = self.prev
        return False


def _act(cfg):
    """get_activation (nlf/activations.py:566-570).  EaseValue (:462-496), w * act(x) + (1 - w) * start_value, is folded
    into the inner activation's `outer` (times w) and `add`; with _ITERATION None (inference) w == 1."""
    if cfg is None:
        cfg = 'identity'
    if isinstance(cfg, str):
        cfg = {'type': cfg}
    scale, add = 1.0, 0.0                      # y_final = scale * inner(x) + add
    while cfg['type'] == 'ease_value':
        w = ease_weight(cfg, _ITERATION)
        add += scale * (1.0 - w) * float(cfg.get('start_value', 0.0))
        scale *= w
        cfg = cfg['activation']
        if isinstance(cfg, str):
            cfg = {'type': cfg}
    t = cfg['type']
    if t not in ACT:
        raise NotImplementedError(f"activation '{t}' is outside the hot-path scope (identity/sigmoid/tanh/ease_value)")
    a = hr_act()
    a.type = ACT[t]
    a.inner = float(cfg.get('inner_fac', 1.0))
    a.shift = float(cfg.get('shift', 0.0))
    a.outer = (float(cfg['fac']) if 'fac' in cfg else float(cfg.get('outer_fac', 1.0))) * scale
    a.add = add
    return a


def _absent_field():
    f = hr_head_field()
    f.offset, f.channels, f.act = -1, 0, _act(None)
    return f


def torch_pow_f32(x, exponent):
    """torch.pow(float32 tensor, python scalar) on CPU: ATen special-cases the exponents 0.5, 2, 3, -0.5, -1, -2
    (pow_tensor_scalar_optimized_kernel) and evaluates the rest as float32 powf."""
    x = np.asarray(x, F32)
    e = float(exponent)
    if e == 0.5:
        return np.sqrt(x).astype(F32)
    if e == 2.0:
        return (x * x).astype(F32)
    if e == 3.0:
        return (x * x * x).astype(F32)
    if e == -0.5:
        return (F32(1.0) / np.sqrt(x)).astype(F32)
    if e == -1.0:
        return (F32(1.0) / x).astype(F32)
    if e == -2.0:
        return (F32(1.0) / (x * x)).astype(F32)
    return np.power(x, F32(e)).astype(F32)


def torch_linspace_f32(start, end, steps):
    """torch.linspace for float32 on CPU: two-sided evaluation around the midpoint."""
    start, end = F32(start), F32(end)
    if steps == 1:
        return np.asarray([start], F32)
    step = F32((end - start) / F32(steps - 1))
    out = np.empty((steps,), F32)
    half = steps // 2
    for i in range(steps):
        out[i] = start + step * F32(i) if i < half else end - step * F32(steps - 1 - i)
    return out


class _MipNerf:
    """Setup-time arithmetic of MIPNeRFContract (nlf/contract.py:113-176), float32 like torch."""

    def __init__(self, c, ds):
        if c.get('use_dataset_bounds', False):
            self.r0 = c.get('contract_start_radius', max(ds['depth_range'][0] * 1.5, 1.0))
            self.r1 = c.get('contract_end_radius', ds['depth_range'][1] * 1.5)
        else:
            self.r0 = c.get('contract_start_radius', 1.0)
            self.r1 = c.get('contract_end_radius', float('inf'))
        self.d0 = c.get('contract_start_distance', self.r0)
        self.d1 = c.get('contract_end_distance', self.r1)
        if 'distance_activation' in c:
            raise NotImplementedError('contract.distance_activation is outside the hot-path scope')

    def contract_distance(self, distance):             # contract.py:160-176, on a 0-dim float32 tensor
        d = F32(distance) / F32(self.d0)
        with np.errstate(divide='ignore'):
            inv = F32(1.0) / np.abs(d)
        inv_end = self.d0 / self.d1
        scale = 1.0 / (1.0 - inv_end)
        t = (inv - F32(inv_end)) * F32(scale)
        out = d / F32(1.0) if np.abs(d) < 1.0 else np.sign(d) * (F32(2.0) - t)
        return F32((F32(out) / F32(2.0)) * F32(2.0))


class _DoNeRF:
    """Setup-time arithmetic of DoNeRFContract (nlf/contract.py:195-240): fac and power are python / numpy doubles there and enter
    the float32 tensor ops as float32 scalars."""

    def __init__(self, c, ds):
        if c.get('use_dataset_bounds', False):
            r0 = c.get('contract_start_radius', max(ds['depth_range'][0] * 1.75, 1.0))
            r1 = c.get('contract_end_radius', ds['depth_range'][1] * 1.5)
        else:
            r0 = c.get('contract_start_radius', None)
            r1 = c.get('contract_end_radius', 10000.0)
        if r0 is None:
            self.power = float(c.get('power', 2.0))
            self.fac = float(np.power(2.0, self.power) / r1)
        else:
            self.fac = 1.0 / r0
            self.power = float(np.log(r1 / r0) / np.log(2.0))
        self.inv_power = 1.0 / self.power
        if 'distance_activation' in c:
            raise NotImplementedError('contract.distance_activation is outside the hot-path scope')

    def contract_distance(self, distance):             # contract.py:232-236, on a 0-dim float32 tensor
        d = F32(distance) * F32(self.fac)
        d = torch_pow_f32(np.abs(d) + F32(1e-8), self.inv_power) * np.sign(d)
        return F32((F32(d) / F32(2.0)) * F32(2.0))


class _Affine:
    """BBoxContract (nlf/contract.py:65-87) and ZDepthContract (:90-111): p -> (p - lo) / size,
    contract_distance(d) = d / fac, inverse = d * fac."""

    def __init__(self, c, ds):
        if c['type'] == 'bbox':
            lo = np.asarray(c.get('bbox_min', [-1.0, -1.0, -1.0]), F32)
            hi = np.asarray(c.get('bbox_max', [1.0, 1.0, 1.0]), F32)
            self.lo, self.size = lo, (hi - lo).astype(F32)
            self.fac = F32(np.mean(np.abs(hi - lo), dtype=F32))
        else:
            if c.get('use_dataset_bounds', False):
                r1 = c.get('contract_end_radius', ds['depth_range'][1])
            else:
                r1 = c.get('contract_end_radius', float('inf'))
            self.fac = F32(r1 / 2.0)
            self.lo, self.size = np.zeros(3, F32), np.full(3, self.fac, F32)

    def contract_distance(self, distance):
        return F32(F32(distance) / self.fac)


# --------------------------------------------------------------------------- the compiler
MLP_PRECISION = {'fp32': 0, 'bf16x3': 1, 'f16x3': 2, 'f16x2': 3, 'auto': 4, 'f16f8': 5, 'f16f8v': 6}


GRID_DTYPE = {'fp32': 0, 'fp16': 1}


def compile_config(cfg, dataset, grid_size, mlp_precision='auto', grid_dtype='fp32', _coarse=False, iteration='inherit'):
    """cfg: `experiment.model` group (dict/Cfg); dataset: {near, far, depth_range,
    num_keyframes, num_frames}; grid_size: [Nx, Ny, Nz] of the uploaded planes.
    iteration: training iteration of the EaseValue / WindowedPE schedules (None: converged; default: whatever an
    enclosing `at_iteration` set, else converged)."""
    if iteration != 'inherit':
        with at_iteration(iteration):
            return compile_config(cfg, dataset, grid_size, mlp_precision, grid_dtype, _coarse)
    if cfg.get('param', {}).get('fn', 'identity') != 'identity':
        raise NotImplementedError("model.param.fn other than 'identity'")
    if cfg['embedding']['type'] != 'ray_point':
        raise NotImplementedError(f"embedding type {cfg['embedding']['type']}")
    hc = hr_config()
    for name in ('f_z_vals', 'f_isect_sigma', 'f_offset_sigma', 'f_point_offset', 'f_color_scale',
                 'f_color_shift', 'f_spatial_flow', 'f_color_scale_global', 'f_color_shift_global'):
        setattr(hc, name, _absent_field())
    stages = list(cfg['embedding']['embeddings'].values())
    types = [s['type'] for s in stages]
    allowed = {'ray_prediction', 'ray_intersect', 'advect_points', 'point_offset', 'add_point_outputs', 'extract_fields',
               'color_transform'}
    for t in types:
        if t not in allowed:
            raise NotImplementedError(f"embedding '{t}' is outside the hot-path scope")
    order = [t for t in types if t in ('ray_prediction', 'ray_intersect', 'advect_points', 'point_offset')]
    if order[:2] != ['ray_prediction', 'ray_intersect'] or order[2:] not in ([], ['point_offset'], ['advect_points'],
                                                                             ['advect_points', 'point_offset']):
        raise NotImplementedError(f'embedding order {order} (expected prediction, intersect[, advect][, offset])')

    # ---- ray_prediction ------------------------------------------------------------
    pred = stages[types.index('ray_prediction')]
    if pred.get('ray_outputs'):
        raise NotImplementedError('ray_outputs')
    if pred.get('rays_name', 'rays') != 'rays':
        raise NotImplementedError('rays_name')
    groups = list(pred['params'].values())
    if len(groups) > HR_MAX_GROUPS:
        raise NotImplementedError(f'more than {HR_MAX_GROUPS} parameter groups')
    hc.n_groups = len(groups)
    mlp_in = 0
    max_col = 0
    for i, g in enumerate(groups):
        pg = hc.groups[i]
        pg.start, pg.end = int(g['start']), int(g['end'])
        max_col = max(max_col, pg.end)
        p = g['param']
        fn = p['fn']
        if fn not in PARAM:
            raise NotImplementedError(f"ray param '{fn}' is outside the hot-path scope")
        if p.get('use_local_param', False):
            raise NotImplementedError('use_local_param')
        pg.fn = PARAM[fn]
        org = p.get('origin', [0.0, 0.0, 0.0])
        for k in range(3):
            pg.origin[k] = float(org[k])
        if fn == 'pluecker':
            if pg.end - pg.start < 6:
                raise ValueError('pluecker needs 6 ray columns')
            pg.a, pg.b = float(p.get('direction_multiplier', 1.0)), float(p.get('moment_multiplier', 1.0))
            n = 6
        elif fn == 'two_plane':
            if pg.end - pg.start < 6:
                raise ValueError('two_plane needs 6 ray columns')
            pg.a, pg.b = float(p.get('near', -1.0)), float(p.get('far', 0.0))
            n = 4
        else:
            n = pg.end - pg.start
            if n > 8:
                raise NotImplementedError('identity param with more than 8 columns')
        pe = g.get('pe')
        if pe is None:
            pg.pe_type = 0
        else:
            if pe['type'] not in ('windowed', 'basic'):
                raise NotImplementedError(f"pe '{pe['type']}' is outside the hot-path scope")
            if 'window_iters' in pe or pe.get('ceil', False):
                raise NotImplementedError('windowed PE with explicit window_iters / ceil is outside the hot-path scope')
            pg.pe_type = PE[pe['type']]
            pg.pe_n_freqs = int(pe['n_freqs'])
            pg.pe_exclude_identity = int(bool(pe.get('exclude_identity', False))) if pe['type'] == 'windowed' else 0
            pg.pe_freq_mult = float(pe.get('freq_multiplier', 2.0))
            pg.pe_base_mult = float(pe.get('base_multiplier', 1.0)) if pe['type'] == 'windowed' else 1.0
            if pe['type'] == 'windowed':
                if pg.pe_n_freqs > 8:
                    raise NotImplementedError('windowed PE with more than 8 frequencies')
                for j, w in enumerate(windowed_pe_weights(pe, _ITERATION)):
                    pg.pe_weight[j] = w
            k = 2 * pg.pe_n_freqs
            n = n * (k if pg.pe_exclude_identity else k + 1)
        mlp_in += n
    hc.mlp_in = mlp_in
    if mlp_in > HR_MAX_MLP_IN:
        raise NotImplementedError(f'MLP input of {mlp_in} features (max {HR_MAX_MLP_IN})')
    net = pred['net']
    if net['type'] not in ('base', 'zero'):
        raise NotImplementedError(f"net '{net['type']}' is outside the hot-path scope (BaseMLP, ZeroMLP)")
    zero_net = net['type'] == 'zero'               # nlf/nets/mlp.py:14-33: all-zero head, the other keys are ignored
    if zero_net:
        net = {'depth': 2, 'hidden_channels': 0}
    for k in ('pe', 'pad_to', 'is_constant', 'zero_before_channel', 'latent_dim'):
        if k in net:
            raise NotImplementedError(f'net.{k}')
    if net.get('layer_activation', 'leaky_relu') != 'leaky_relu' or net.get('activation', 'identity') != 'identity':
        raise NotImplementedError('MLP activations other than leaky_relu / identity')
    if not net.get('bias', True):
        raise NotImplementedError('bias-free MLP')
    D = int(net['depth']) - 2                       # ray.py:283-285
    hc.mlp_layers = 0 if zero_net else D + 2
    hc.mlp_hidden = int(net['hidden_channels'])
    mask = 0
    for s in net.get('skips', []):
        if 0 < s < D + 2:
            mask |= 1 << int(s)
        elif s == 0:
            raise NotImplementedError('skip connection into layer 0')
    hc.mlp_skip_mask = mask
    hc.leaky_slope = 0.01
    Z = int(pred['z_channels'])
    if Z > HR_KERNEL_MAX_Z:
        raise NotImplementedError(f'z_channels {Z} > {HR_KERNEL_MAX_Z}')
    hc.z_channels = Z
    heads = {}
    off = 0
    for name, o in pred['outputs'].items():
        f = hr_head_field()
        f.offset, f.channels, f.act = off, int(o['channels']), _act(o.get('activation'))
        heads[name] = f
        off += f.channels
    hc.preds_per_z = off
    if 'z_vals' not in heads:
        raise NotImplementedError('model without a z_vals head')
    hc.f_z_vals = heads['z_vals']
    # ExtractFieldsEmbedding (embedding/point.py:236-244): the colour net only sees the listed fields
    seen = set(stages[types.index('extract_fields')]['fields']) if 'extract_fields' in types else None
    has = lambda k: k in heads and (seen is None or k in seen)
    if has('color_scale'):                           # tensorf_no_sample.py:222-225 (color_shift is read unguarded)
        if not has('color_shift'):
            raise ValueError('color_scale without color_shift')
        hc.f_color_scale, hc.f_color_shift = heads['color_scale'], heads['color_shift']
    elif has('color_transform'):
        raise NotImplementedError("head 'color_transform' is outside the hot-path scope")
    if has('color_scale_global'):                    # tensorf_no_sample.py:240-241
        if not has('color_shift_global'):
            raise ValueError('color_scale_global without color_shift_global')
        hc.f_color_scale_global, hc.f_color_shift_global = heads['color_scale_global'], heads['color_shift_global']
    elif has('color_transform_global'):              # tensorf_no_sample.py:242-243 -> transform_color_one: a 3x3 per ray from the head
        if not has('color_shift_global'):
            raise ValueError('color_transform_global without color_shift_global')
        if heads['color_transform_global'].channels != 9:
            raise ValueError('color_transform_global needs 9 channels')
        # carried in the scale field: 9 channels = the row-major matrix (hr_write_pixel, csrc/sample_core.inc)
        hc.f_color_scale_global, hc.f_color_shift_global = heads['color_transform_global'], heads['color_shift_global']
    for k in ('f_color_scale', 'f_color_shift', 'f_color_scale_global', 'f_color_shift_global'):
        ok = (3, 9) if (k == 'f_color_scale_global' and not has('color_scale_global')) else (3,)
        if getattr(hc, k).offset >= 0 and getattr(hc, k).channels not in ok:
            raise ValueError(f'{k[2:]} needs 3 channels')
    if has('weights_shift'):
        raise NotImplementedError("head 'weights_shift' is outside the hot-path scope")
    # ColorTransformEmbedding (embedding/point.py:558-602): a no-op unless dataset.val_all
    hc.color_table_t_act, hc.color_table_s_act = _act(None), _act(None)
    if 'color_transform' in types and dataset.get('val_all', False):
        ct = stages[types.index('color_transform')]
        tf, sf = ct.get('out_transform_field', 'color_transform_global'), ct.get('out_shift_field', 'color_shift_global')
        if (tf, sf) != ('color_transform_global', 'color_shift_global'):
            raise NotImplementedError('color_transform with renamed output fields')
        if 'color_shift_global' in heads:
            raise NotImplementedError('color_transform next to a color_shift_global head')
        vis = lambda k: seen is None or k in seen
        # tensorf_no_sample.py:240-243: a color_scale_global head takes precedence over the transform
        if vis(tf) and vis(sf) and hc.f_color_scale_global.offset < 0:
            hc.color_table_views = int(dataset['total_images_per_frame'])
            hc.color_table_t_act = _act(ct.get('transform_activation'))
            hc.color_table_s_act = _act(ct.get('shift_activation'))

    # ---- ray_intersect --------------------------------------------------------------
    st = stages[types.index('ray_intersect')]
    if int(st['z_channels']) != Z:
        raise ValueError('ray_prediction and ray_intersect disagree on z_channels')
    ic = st['intersect']
    t = ic['type']
    if t not in ISECT:
        raise NotImplementedError(f"intersect '{t}' is outside the hot-path scope (SURVEY 8f-1)")
    hc.isect_type = ISECT[t]
    if hc.f_z_vals.channels != ISECT_Z_CHANNELS[hc.isect_type]:
        # the reference reshapes z_vals to (B, Z, n) and fails (or silently mis-strides) otherwise
        raise ValueError(f"intersect '{t}' needs {ISECT_Z_CHANNELS[hc.isect_type]} z_vals channels, "
                         f"the head has {hc.f_z_vals.channels}")
    for k in ('weight_fn', 'sort_outputs', 'dropout', 'normalize', 'residual_z', 'residual_distance', 'clamp',
              'use_local_prediction', 'flip_axes', 'use_disparity', 'max_axis'):
        if ic.get(k):
            raise NotImplementedError(f'intersect.{k}')
    if ic.get('num_repeat', 1) != 1:
        raise NotImplementedError('intersect.num_repeat')
    if 'mask' in ic:                              # base.py:104-108,197-198; inference runs at iter 1e7
        hc.isect_mask_off = int((10_000_000 if _ITERATION is None else _ITERATION) > ic['mask'].get('stop_iters', float('inf')))
    udb = ic.get('use_dataset_bounds', False)
    org = ic.get('origin', [0.0, 0.0, 0.0])
    for k in range(3):
        hc.isect_origin[k] = float(org[k])
    hc.near = float(ic['near']) if 'near' in ic else (float(dataset['near']) if udb else 0.0)   # base.py:88-94
    hc.far = float(ic.get('far', float('inf')))
    hc.z_act = _act(ic.get('activation'))
    hc.sort = int(bool(ic.get('sort', False)))
    if ic.get('use_sigma', False):
        fld = ic.get('in_density_field', 'sigma')
        if fld in heads:
            if heads[fld].channels != 1:
                raise NotImplementedError('intersect density field with more than one channel')
            hc.f_isect_sigma = heads[fld]
    contract = None
    if 'contract' in ic:
        ct = ic['contract']['type']
        if ct not in CONTRACT:
            raise NotImplementedError(f"contract '{ct}' is outside the hot-path scope")
        if 'stop_iters' in ic['contract']:
            raise NotImplementedError('contract.stop_iters')
        hc.contract_type = CONTRACT[ct]
        hc.contract_samples = int(bool(ic['contract'].get('contract_samples', False)))
        if ct in ('bbox', 'z_depth'):
            contract = _Affine(ic['contract'], dataset)
            for k in range(3):
                hc.c_aff_min[k], hc.c_aff_size[k] = float(contract.lo[k]), float(contract.size[k])
            hc.c_aff_fac = float(contract.fac)
        elif ct == 'mipnerf':
            contract = _MipNerf(ic['contract'], dataset)
            hc.c_r0 = float(contract.r0)
            inv_end = contract.r0 / contract.r1
            hc.c_r_inv_end = float(inv_end)
            hc.c_r_scale = float(1.0 / (1.0 - inv_end))
            hc.c_d0 = float(contract.d0)
            inv_end = contract.d0 / contract.d1
            hc.c_d_inv_end = float(inv_end)
            hc.c_d_scale = float(1.0 / (1.0 - inv_end))
        elif ct == 'donerf':
            contract = _DoNeRF(ic['contract'], dataset)
            hc.c_pow_fac, hc.c_pow_power, hc.c_pow_inv_power = float(contract.fac), float(contract.power), float(contract.inv_power)
        elif hc.contract_samples:
            hc.contract_samples = 0          # IdentityContract.inverse_contract_distance is the identity
    cdist = contract.contract_distance if (hc.contract_samples and contract is not None) else (lambda v: F32(v))
    if t == 'voxel_grid':                     # voxel.py:19-70: Z/3 axis planes per axis, sample k -> axis k % 3
        if Z % 3:
            raise ValueError('voxel_grid needs z_channels divisible by 3')
        nz = Z // 3
        fac = ic.get('fac', 1.0)
        if udb:
            if 'bbox_min' not in dataset and not ('initial' in ic and 'end' in ic):
                raise ValueError('voxel_grid with use_dataset_bounds reads dataset.bbox_min / bbox_max')
            initial = ic['initial'] if 'initial' in ic else (np.asarray(dataset['bbox_min'], F32) * F32(fac))
            end = ic['end'] if 'end' in ic else (np.asarray(dataset['bbox_max'], F32) * F32(fac))
        else:
            initial, end = ic.get('initial', [0.0, 0.0, 0.0]), ic.get('end', [1.0, 1.0, 1.0])
        cols = [torch_linspace_f32(cdist(F32(initial[d])), cdist(F32(end[d])), nz) for d in range(3)]
        for j in range(nz):
            for d in range(3):
                hc.samples[3 * j + d] = float(cols[d][j])
        for d in range(3):
            if 'z_scale' in ic:
                zs = F32(ic['z_scale'][d])
            else:
                zs = np.abs(cols[d][1] - cols[d][0]) if nz > 1 else F32(1.0)
            hc.voxel_scale[d] = float(zs) if zs != 0 else 1.0
        hc.z_scale = 1.0
        hc.isect_outward = int(bool(ic.get('outward_facing', False)))
        samples = None
    elif t == 'deformable_voxel_grid':        # voxel.py:115-176: one plane family per start normal
        if udb:
            raise NotImplementedError('deformable_voxel_grid with use_dataset_bounds (reads the dataset point cloud)')
        normals = ic.get('start_normal', [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
        na = len(normals)
        if not 1 <= na <= 3 or Z % na:
            raise ValueError('deformable_voxel_grid needs 1..3 start normals dividing z_channels')
        hc.dvg_axes = na
        for a_ in range(na):
            for i in range(3):
                hc.dvg_normals[3 * a_ + i] = float(normals[a_][i])
        hc.dvg_normal_scale = float(ic.get('normal_scale_factor', 0.1))
        nz = Z // na
        ini, en = ic.get('initial', [0.0, 0.0, 0.0]), ic.get('end', [1.0, 1.0, 1.0])
        cols = [torch_linspace_f32(cdist(F32(ini[d])), cdist(F32(en[d])), nz) for d in range(na)]
        flat = np.stack(cols, -1).reshape(-1).astype(F32)       # torch.stack(samples, -1).view(-1, 1)
        for k in range(Z):
            hc.samples[k] = float(flat[k])
        if 'z_scale' in ic:
            if na != 1:
                raise NotImplementedError('deformable_voxel_grid.z_scale with more than one axis (the reference cannot broadcast it)')
            zs = F32(ic['z_scale'][0])
        else:
            # voxel.py:171-172 takes |samples[1] - samples[0]| of the FLATTENED (Z, 1) samples
            zs = np.abs(flat[1] - flat[0]) if nz > 1 else F32(1.0)
        hc.z_scale = float(zs) if zs != 0 else 1.0
        samples = None
    elif t == 'z_plane':                      # z.py:25-39
        if udb:
            initial, end = F32(-dataset['near']), F32(-dataset['far'])
        else:
            initial, end = F32(ic.get('initial', 0.0)), F32(ic.get('end', 1.0))
    elif t in ('sphere', 'cylinder'):         # primitive.py:185-215 / 370-400
        if udb:
            initial = F32(ic['initial']) if 'initial' in ic else F32(dataset['near'] * 1.5)
            end = F32(ic['end']) if 'end' in ic else F32(dataset['far'] * 1.5)
        else:
            initial, end = F32(ic.get('initial', 0.0)), F32(ic.get('end', 1.0))
        hc.origin_scale = float(ic.get('origin_scale_factor', 0.0))
        oi = ic.get('origin_initial', [1.0, 1.0, 1.0])
        for k in range(3):
            hc.origin_initial[k] = float(oi[k])
    elif t in ('sphere_new', 'cylinder_new'):   # primitive.py:256-303 / 441-488
        if udb:
            if ic['outward_facing']:
                initial = F32(ic['initial']) if 'initial' in ic else F32(dataset['near'] * 1.5)
            else:
                initial = F32(ic['initial']) if 'initial' in ic else F32(-dataset['far'] * 1.5)
            end = F32(ic['end']) if 'end' in ic else F32(dataset['far'] * 1.5)
        else:
            initial, end = F32(ic.get('initial', 0.0)), F32(ic.get('end', 1.0))
        hc.origin_scale = float(ic.get('origin_scale_factor', 0.0))
        hc.resize_scale = float(ic.get('resize_scale_factor', 0.0))
        ri = ic.get('resize_initial', [1.0, 1.0, 1.0])
        for k in range(3):
            hc.resize_initial[k] = float(ri[k])
    else:                                     # euclidean_distance_unified, primitive.py:131-160
        if udb:
            initial = F32(ic['initial']) if 'initial' in ic else F32(-dataset['far'])
            end = F32(ic['end']) if 'end' in ic else F32(dataset['far'])
        else:
            initial, end = F32(ic.get('initial', 0.0)), F32(ic.get('end', 1.0))
    if t not in ('voxel_grid', 'deformable_voxel_grid'):
        samples = torch_linspace_f32(cdist(initial), cdist(end), Z)
        for k in range(Z):
            hc.samples[k] = float(samples[k])
        if Z > 1:
            if 'z_scale' in ic:
                zs = F32(ic['z_scale'])
            elif 'num_samples_for_scale' in ic and t == 'z_plane':       # z.py:63-65
                zs = np.abs(samples[1] - samples[0]) * F32(Z / float(ic['num_samples_for_scale']))
            else:
                zs = np.abs(samples[1] - samples[0])
        else:
            zs = F32(ic.get('z_scale', 1.0))
        hc.z_scale = float(zs)

    # ---- advect / offset --------------------------------------------------------------
    if 'advect_points' in types:
        ad = stages[types.index('advect_points')]
        if ad.get('use_angular_flow', False):
            raise NotImplementedError('angular flow')
        hc.advect = 1
        hc.use_spatial_flow = int(bool(ad.get('use_spatial_flow', False)))
        K, Fr = int(dataset['num_keyframes']), int(dataset['num_frames'])
        if K > 0:
            fac = K * (Fr - 1) / Fr           # flow_utils.py:18-19
            hc.flow_fac = float(fac)
            hc.flow_inv_fac = float(1.0 / fac)
            hc.flow_kmax = float(K - 1.0)
        hc.flow_act = _act(ad.get('spatial_flow_activation'))
        if hc.use_spatial_flow:
            if 'spatial_flow' not in heads:
                raise ValueError('advect_points needs a spatial_flow head')
            hc.f_spatial_flow = heads['spatial_flow']
    else:
        hc.flow_act = _act(None)
    if 'point_offset' in types:
        po = stages[types.index('point_offset')]
        for k in ('dropout', 'in_points_field', 'out_points_field', 'in_offset_field'):
            if k in po and po[k] not in (None, 'points', 'point_offset'):
                raise NotImplementedError(f'point_offset.{k}')
        hc.point_offset = 1
        hc.offset_act = _act(po.get('activation'))
        if 'point_offset' not in heads:
            raise ValueError('point_offset needs a point_offset head')
        hc.f_point_offset = heads['point_offset']
        fld = po.get('in_density_field', 'sigma')
        if po.get('use_sigma', True) and fld in heads:
            if heads[fld].channels != 1:
                raise NotImplementedError('offset density field with more than one channel')
            hc.f_offset_sigma = heads[fld]
    else:
        hc.offset_act = _act(None)

    # ---- colour net ---------------------------------------------------------------------
    if cfg['color']['type'] != 'base':
        raise NotImplementedError(f"color model {cfg['color']['type']}")
    n = cfg['color']['net']
    if n['type'] not in ('tensor_vm_split_no_sample', 'tensor_vm_split_time'):
        raise NotImplementedError(f"colour net '{n['type']}' is outside the hot-path scope")
    hc.video = int(n['type'] == 'tensor_vm_split_time')
    if 'filter' in n:
        raise NotImplementedError('net.filter (apply_filter_weights)')
    aabb = np.asarray(n['aabb'] if _AABB is None else _AABB, F32).reshape(2, 3)
    for k in range(3):
        hc.aabb[k], hc.aabb[3 + k] = float(aabb[0][k]), float(aabb[1][k])
    inv = F32(2.0) / (aabb[1] - aabb[0])      # tensorf_base.py:296-297
    for k in range(3):
        hc.inv_size[k] = float(inv[k])
        hc.grid[k] = int(grid_size[k])
    nd, na = list(n.get('n_lamb_sigma', [8, 8, 8])), list(n.get('n_lamb_sh', [24, 24, 24]))
    for k in range(3):
        hc.n_den[k], hc.n_app[k] = int(nd[k]), int(na[k])
    shading = n.get('shadingMode', 'MLP_PE')
    if shading not in SHADING:
        raise NotImplementedError(f"shadingMode '{shading}' is outside the hot-path scope (RGB, SH)")
    hc.shading = SHADING[shading]
    hc.app_dim = int(n.get('data_dim_color', 27))
    hc.distance_scale = float(n.get('distance_scale', 25))
    hc.weight_thresh = float(n.get('rm_weight_mask_thre', 0.0001))
    act = n.get('fea2denseAct', 'softplus')
    if act not in DENSITY:
        raise NotImplementedError(f'fea2denseAct {act}')
    hc.density_act = DENSITY[act]
    hc.density_shift = float(n.get('density_shift', -10.0))
    hc.white_bg = int(bool(n.get('white_bg', 0)) and not bool(n.get('black_bg', 0)))
    hc.ray_dim = 8 if (hc.video or max_col > 6 or hc.advect or hc.color_table_views > 0) else 6   # camera id = rays[..., -2]
    if hc.video:
        if n.get('densityMode', 'Density') != 'Density':
            raise NotImplementedError(f"densityMode {n.get('densityMode')}")
        K, Fr = int(dataset['num_keyframes']), int(dataset['num_frames'])
        hc.num_keyframes = K
        hc.time_scale = float((Fr - 1) / Fr)  # tensorf_dynamic.py:58-59
        hc.time_offset = float(0.5 / K)
        if not hc.advect and not _coarse:
            raise NotImplementedError('video net without an advect_points stage (base_times)')
    # GEMM arithmetic of the MLP.  'auto' (HR_MLP_AUTO) lets the library choose when the kernel supports the width: 'f16x3' -- three
    # fp16 MFMA products of hi/lo split operands (11 + 11 mantissa bits, weights pre-scaled by an exact power of two), fp32 accumulate,
    # raw head within 1e-6 of the exact fp32 chain (bf16x3: 7e-6, same cost), which is what keeps the reference's threshold decisions
    # (`dist <= near`, intersect/base.py:194-203) from flipping -- wherever hr_model_finalize's activation-range calibration shows the
    # MLP's activations to stay below 65504 / 8 (fp16 halves saturate at 65504; the reference's BaseMLP is fp32, nlf/nets/mlp.py:127-172),
    # and 'bf16x3' (fp32 exponent range) otherwise.  Round 5: where the model allows it (a plain ray MLP, <= 64 samples per ray) 'auto' is 'f16f8v' --
    # the f16 + fp8 arithmetic (two thirds of f16x3's matrix-pipe time) with every ray that has a comparison within 2.5e-6 of the scene's extent of
    # flipping rendered again with the f16x3 tiles by a second, list-driven pass on the device (include/hyperreel_hip.h, HR_MLP_F16F8V).  A forced 'f16x3' / 'f16x2' that fails the same test is refused by name
    # (HR_E_RANGE).  'fp32' is the exact fp32 MFMA; other hidden widths only have that.
    if mlp_precision == 'auto' or hc.mlp_layers == 0:
        mlp_precision = 'auto' if hc.mlp_hidden == 256 else 'fp32'
    if mlp_precision in ('bf16x3', 'f16x3', 'f16x2', 'f16f8', 'f16f8v') and hc.mlp_hidden != 256:
        raise NotImplementedError(f'{mlp_precision} MLP needs hidden_channels == 256')
    hc.mlp_precision = MLP_PRECISION[mlp_precision]
    if grid_dtype not in GRID_DTYPE:
        raise ValueError(f"grid_dtype must be one of {sorted(GRID_DTYPE)} (got {grid_dtype!r})")
    hc.grid_dtype = GRID_DTYPE[grid_dtype]
    return hc


PIN = {'points': 0, 'viewdirs': 1, 'origins': 2, 'times': 3}


def is_cascade(cfg):
    return any(e.get('type') == 'point_prediction' for e in cfg.get('embedding', {}).get('embeddings', {}).values())


def compile_cascade(cfg, dataset, grid_size, mlp_precision='auto', grid_dtype='fp32'):
    """point_prediction cascades (embedding/point.py:39-218): ray_prediction -> ray_intersect -> point_prediction
    -> ray_intersect -> ...  Returns (coarse, fine) hr_configs for hr_model_create_cascade.  The point MLP is
    described to `compile_config` as a ray_prediction over the columns of its input row."""
    import copy
    items = list(cfg['embedding']['embeddings'].items())
    types = [e['type'] for _, e in items]
    ip = types.index('point_prediction')
    if types[:ip] != ['ray_prediction', 'ray_intersect'] or types[ip + 1:ip + 2] != ['ray_intersect'] \
            or 'point_prediction' in types[ip + 1:] or 'ray_prediction' in types[ip + 1:]:
        raise NotImplementedError(f'cascade layout {types} (expected ray_prediction, ray_intersect, point_prediction, '
                                  f'ray_intersect, ...)')
    pp = items[ip][1]
    for k in ('filter',):
        if pp.get(k):
            raise NotImplementedError(f'point_prediction.{k}')
    if pp.get('rays_name', 'rays') != 'rays' or pp.get('points_name', 'points') != 'points':
        raise NotImplementedError('point_prediction with renamed rays / points')
    if any(o.get('residual', False) for o in pp['outputs'].values()):
        raise NotImplementedError('residual point_prediction outputs')
    in_z, out_z = int(pp.get('in_z_channels', 1)), int(pp.get('out_z_channels', 1))
    Z0 = int(items[0][1]['z_channels'])
    if in_z != Z0 or out_z % in_z:
        raise ValueError(f'point_prediction in_z_channels {in_z} / out_z_channels {out_z} vs {Z0} coarse samples')
    kinds, dims = [], []
    for name, n in pp['inputs'].items():
        if name not in PIN:
            raise NotImplementedError(f"point_prediction input '{name}'")
        full = 1 if name == 'times' else 3
        n = int(n)
        if not 1 <= n <= full:
            raise ValueError(f"point_prediction input '{name}': {n} columns")
        # viewdirs / origins / times are appended whole whatever the declared width (point.py:147-153)
        kinds.append(PIN[name])
        dims.append(n if name == 'points' else full)
    if len(kinds) > 4:
        raise NotImplementedError('more than 4 point_prediction inputs')
    row_dim = sum(dims)
    if row_dim > 8:
        raise NotImplementedError('point_prediction rows wider than 8 columns')
    # coarse level: the first two stages; the colour net rides along unused
    c0 = copy.deepcopy(cfg)
    c0['embedding']['embeddings'] = type(cfg['embedding']['embeddings'])(items[:2])
    hc0 = compile_config(c0, dataset, grid_size, mlp_precision, grid_dtype, _coarse=True)
    # fine level: the point MLP posed as a ray_prediction over the row's columns + everything after it
    c1 = copy.deepcopy(cfg)
    synth = type(pp)({'type': 'ray_prediction', 'params': copy.deepcopy(pp['params']), 'net': copy.deepcopy(pp['net']),
                      'z_channels': out_z, 'outputs': copy.deepcopy(pp['outputs'])})
    for g in synth['params'].values():
        if g['param'].get('fn', 'identity') != 'identity':
            raise NotImplementedError('point_prediction params other than identity')
        if g['end'] > row_dim:
            raise ValueError('point_prediction param reads beyond the input row')
    c1['embedding']['embeddings'] = type(cfg['embedding']['embeddings'])([(items[ip][0], synth)] + items[ip + 1:])
    hc1 = compile_config(c1, dataset, grid_size, mlp_precision, grid_dtype)
    hc1.ray_dim = hc0.ray_dim = max(hc0.ray_dim, 8 if (hc1.video or hc1.advect or hc1.color_table_views > 0) else 6)
    hc1.casc_in_z, hc1.casc_row_dim, hc1.casc_n_inputs = in_z, row_dim, len(kinds)
    for i, (k, d) in enumerate(zip(kinds, dims)):
        hc1.casc_input_kind[i], hc1.casc_input_dim[i] = k, d
    return hc0, hc1


def compile_model(cfg, dataset, grid_size, mlp_precision='auto', grid_dtype='fp32', iteration=None, aabb=None):
    """-> (coarse hr_config or None, hr_config of the rendering level): cascade-aware front door.
    iteration: training iteration of the activation / PE schedules (None: converged, what render and test use).
    aabb: the colour net's `aabb` buffer ((2, 3); None: the YAML's `color.net.aabb`)."""
    with at_iteration(iteration, aabb):
        if is_cascade(cfg):
            return compile_cascade(cfg, dataset, grid_size, mlp_precision, grid_dtype)
        return None, compile_config(cfg, dataset, grid_size, mlp_precision, grid_dtype)


def live_head_columns(hc):
    """Per-sample head columns the path reads (mirror of analyse_live_columns in csrc/api.hip).
    The library drops the others from the last Linear; `hr_render_fields` reports them as 0."""
    live = [False] * hc.preds_per_z

    def mark(f, first, count):
        if f.offset >= 0:
            for i in range(first, first + count):
                live[f.offset + i] = True

    t = hc.isect_type
    if t in (ISECT['sphere'], ISECT['cylinder']):
        mark(hc.f_z_vals, 3, 1)
        if hc.origin_scale != 0.0:
            mark(hc.f_z_vals, 0, 3)
    elif t == ISECT['deformable_voxel_grid']:
        mark(hc.f_z_vals, 3, 1)
        if hc.dvg_normal_scale != 0.0:
            mark(hc.f_z_vals, 0, 3)
    elif t in (ISECT['sphere_new'], ISECT['cylinder_new']):
        mark(hc.f_z_vals, 6, 2)
        if hc.resize_scale != 0.0 or hc.origin_scale != 0.0:   # kept contiguous up to channel 7
            mark(hc.f_z_vals, 3, 3)
        if hc.origin_scale != 0.0:
            mark(hc.f_z_vals, 0, 3)
    else:
        mark(hc.f_z_vals, 0, 1)
    mark(hc.f_isect_sigma, 0, 1)
    if hc.point_offset:
        mark(hc.f_point_offset, 0, 3)
        mark(hc.f_offset_sigma, 0, 1)
    mark(hc.f_color_scale, 0, 3)
    mark(hc.f_color_shift, 0, 3)
    mark(hc.f_color_scale_global, 0, 3)
    mark(hc.f_color_shift_global, 0, 3)
    if hc.advect and hc.use_spatial_flow:
        mark(hc.f_spatial_flow, 0, 3)
    return live


def upload_names(hc, coarse=None):
    """[(ABI tensor name, reference state_dict key suffix)] for hr_model_upload.  Keys carry {idx} (the
    ray_prediction stage), {pp_idx} (the point_prediction stage of a cascade) and {ct_idx} (color_transform)."""
    names = []

    def mlp(prefix, L, where):
        emb = 'embedding_model.embeddings.{' + where + '}.net.layers.'
        for i in range(L):
            mid = '.0' if i < L - 1 else ''    # Sequential(Linear, act) vs bare Linear (mlp.py:149-154)
            names.append((f'{prefix}.{i}.weight', f'{emb}{i}{mid}.weight'))
            names.append((f'{prefix}.{i}.bias', f'{emb}{i}{mid}.bias'))

    if coarse is None:
        mlp('mlp', hc.mlp_layers, 'idx')
    else:                                      # hr_model_create_cascade: coarse ray MLP, then the point MLP
        mlp('mlp', coarse.mlp_layers, 'idx')
        mlp('mlp1', hc.mlp_layers, 'pp_idx')
    kinds = ('plane_space', 'plane_time') if hc.video else ('plane', 'line')
    for what in ('density', 'app'):
        for kind in kinds:
            for j in range(3):
                names.append((f'{what}_{kind}.{j}', f'color_model.net.{what}_{kind}.{j}'))
    names.append(('basis_mat.weight', 'color_model.net.basis_mat.weight'))
    if hc.color_table_views > 0:
        names.append(('color_embedding', 'embedding_model.embeddings.{ct_idx}.color_embedding'))
    return names
