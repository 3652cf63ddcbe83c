"""CPU oracle for the HyperReel forward-render hot path (numpy, fp32).

TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / reported CPU baseline.  The product path
(`hyperreel_amd`) never imports anything under `oracle/` and raises if its HIP
library is missing.

It is a restatement, in plain numpy float32, of what the reference computes
for `render_fn(rays)['rgb']` (paths relative to /root/reference):

    nlf/rendering.py:72-77,100-150      RenderLightfield.forward / render_chunked
    nlf/models/models.py:131-138        LightfieldModel.embed / forward
    nlf/embedding/embedding.py:100-117  RayPointEmbedding.forward
    nlf/embedding/ray.py:316-347        RayPredictionEmbedding.forward
    nlf/param.py:87-115,244-253         TwoPlaneParam / PlueckerParam
    nlf/pe.py:210-221,53-66             WindowedPE / BasicPE
    nlf/nets/mlp.py:159-172             BaseMLP.forward
    nlf/activations.py                  Identity/Sigmoid/Tanh/EaseValue/LeakyReLU
    nlf/intersect/base.py:128-259       Intersect.process_z_vals / forward
    nlf/intersect/z.py:16-97            IntersectZPlane
    nlf/intersect/primitive.py:181-253,366-438   IntersectCylinderOld / IntersectSphereOld
    utils/intersect_utils.py:12-16,45-150        sort_z, intersect_sphere/cylinder/axis_plane
    nlf/contract.py:43-50,113-192       MIPNeRFContract (+ IdentityContract)
    nlf/embedding/point.py:371-396,780-831,857-869,236-244   offset / advect / add outputs / extract
    utils/flow_utils.py:10-35           get_base_time
    nlf/nets/tensorf_no_sample.py:47-280         TensorVMNoSample
    nlf/nets/tensorf_dynamic.py:287-371,615-839  TensorVMKeyframeTime
    nlf/nets/tensorf_base.py:308-309,349-353     normalize_coord / valid_mask
    utils/tensorf_utils.py:242-273,334-343,403-404   raw2alpha, colour transforms, SH/RGB/Density render
    utils/sh_utils.py:94-119            eval_sh_bases (degree 2)

Parity pin: the reference ships no golden vectors for this path (SURVEY.md
section 8c), so this oracle is pinned against outputs of the reference itself,
run in the authoring container through `oracle/refgen/ref_shim.py`; the
fixtures and the generating script are committed under `tests/golden/` and
`oracle/refgen/make_golden.py`, and `tests/test_oracle_golden.py` re-checks
them on every run.

Inference-time simplifications (all verified against the reference through
the goldens): `set_iter(10_000_000)` makes every EaseValue / WindowedPE weight
exactly 1 (nlf/__init__.py:582-583, activations.py:476-483, pe.py:184-193);
`self.training` is False; `apply_filter_weights` is off and the alpha-mask
branch is dead (`and False`, tensorf_no_sample.py:171).
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- helpers
def _f(x):
    return np.asarray(x, dtype=F32)


def torch_linspace(start, end, steps):
    """torch.linspace on CPU for float32 (symmetric two-sided evaluation)."""
    start, end = F32(start), F32(end)
    if steps == 0:
        return np.zeros((0,), F32)
    if steps == 1:
        return np.array([start], F32)
    step = F32((end - start) / F32(steps - 1))
    i = np.arange(steps)
    half = steps // 2
    lo = start + step * i.astype(F32)
    hi = end - step * (steps - 1 - i).astype(F32)
    return np.where(i < half, lo, hi).astype(F32)


def sigmoid(x):
    with np.errstate(over='ignore'):
        return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


# Training iteration the EaseValue / WindowedPE schedules are evaluated at (None: converged -- set_iter(1e7), what the
# reference uses for render and test, nlf/__init__.py:582-583).  Set by HyperReelOracle(..., iteration=i) while it builds.
ITERATION = None


def windowed_pe_weights(pe, iteration):
    """WindowedPE.weight(j + window_identity), j < n_freqs (nlf/pe.py:166-208)."""
    n = int(pe['n_freqs'])
    max_freq_iter = float(pe.get('max_freq_iter', 0))
    wait = float(pe.get('wait_iters', 0))
    wid = 1 if pe.get('window_identity', False) else 0
    if iteration is None or n == 0 or max_freq_iter == 0:
        return [1.0] * n
    after = max_freq_iter / n
    if wid:
        win = [(wait, after + wait)] + [(after * i + wait, after * (i + 1) + wait) for i in range(1, n + 1)]
        max_freq_iter = (n + 1) * after
    else:
        win = [(after * i + wait, after * (i + 1) + wait) for i in range(n)]
    out = []
    for j in range(wid, n + wid):
        cur = iteration - wait
        if cur < 0:
            out.append(0.0)
        elif iteration > max_freq_iter:
            out.append(1.0)
        elif win[j][1] - win[j][0] == 0:
            out.append(1.0 if iteration >= win[j][0] else 0.0)
        else:
            a = (cur - win[j][0]) / float(win[j][1] - win[j][0])
            out.append(float((1.0 - np.cos(np.pi * np.clip(a, 0.0, 1.0))) / 2))
    return out


class Act:
    """nlf/activations.py get_activation().  EaseValue (activations.py:462-496) returns
    w * act(x) + (1 - w) * start_value with w = 1 once cur_iter = iteration - wait_iters has reached window_iters."""

    def __init__(self, cfg):
        if cfg is None:
            cfg = 'identity'
        if isinstance(cfg, str):
            cfg = {'type': cfg}
        self.ease = []                                # outermost first: (w, start_value)
        while cfg['type'] == 'ease_value':            # activations.py:462-496
            w = 1.0
            if ITERATION is not None:
                cur = ITERATION - cfg.get('wait_iters', 0.0)
                window = cfg.get('window_iters', 0.0)
                if cur >= window:
                    w = 1.0
                elif window == 0:
                    w = 0.0
                else:
                    w = min(max(float(cur) / window, 0.0), 1.0)
            self.ease.append((w, float(cfg.get('start_value', 0.0))))
            cfg = cfg['activation']
            if isinstance(cfg, str):
                cfg = {'type': cfg}
        self.type = cfg['type']
        self.inner = F32(cfg.get('inner_fac', 1.0))
        self.outer = F32(cfg.get('outer_fac', 1.0))
        self.shift = F32(cfg.get('shift', 0.0))
        if 'fac' in cfg:
            self.outer = F32(cfg['fac'])
        if self.type not in ('identity', 'sigmoid', 'tanh'):
            raise NotImplementedError(f'activation {self.type} is outside the hot-path scope')

    def __call__(self, x):
        y = x * self.inner + self.shift
        if self.type == 'sigmoid':                    # activations.py:53-69
            y = sigmoid(y)
        elif self.type == 'tanh':                     # activations.py:121-137
            y = np.tanh(y)
        y = (y * self.outer).astype(F32)              # identity: activations.py:163-178
        for w, sv in reversed(self.ease):             # ease_out, innermost first; w == 1 returns `out` unchanged (:481)
            if w != 1.0:
                y = (F32(w) * y + F32((1 - w) * sv)).astype(F32)
        return y


# --------------------------------------------------------------------------- contraction
class MipNerfContract:
    """nlf/contract.py:113-192."""

    def __init__(self, cfg, dataset):
        self.contract_samples = bool(cfg.get('contract_samples', False))
        if cfg.get('use_dataset_bounds', False):
            self.r0 = cfg.get('contract_start_radius', max(dataset['depth_range'][0] * 1.5, 1.0))
            self.r1 = cfg.get('contract_end_radius', dataset['depth_range'][1] * 1.5)
        else:
            self.r0 = cfg.get('contract_start_radius', 1.0)
            self.r1 = cfg.get('contract_end_radius', float('inf'))
        self.d0 = cfg.get('contract_start_distance', self.r0)
        self.d1 = cfg.get('contract_end_distance', self.r1)
        if 'distance_activation' in cfg:
            raise NotImplementedError('distance_activation is outside the hot-path scope')

    def inverse_contract_distance(self, distance):     # contract.py:143-158
        inv_end = self.d0 / self.d1                    # python floats, as in the reference
        scale = 1.0 / (1.0 - inv_end)
        distance = (distance / F32(2.0)) * F32(2.0)    # identity activation
        distance = np.clip(distance, F32(-2.0), F32(2.0))
        t = F32(2.0) - np.abs(distance)
        inv = t / F32(scale) + F32(inv_end)
        with np.errstate(divide='ignore'):
            out = np.where(np.abs(distance) < 1, distance, np.sign(distance) * (F32(1.0) / inv))
        return (out * F32(self.d0)).astype(F32)

    def contract_distance(self, distance):             # contract.py:160-176
        distance = _f(distance) / F32(self.d0)
        with np.errstate(divide='ignore'):
            inv = F32(1.0) / np.abs(distance)
        inv_end = self.d0 / self.d1
        scale = 1.0 / (1.0 - inv_end)
        t = (inv - F32(inv_end)) * F32(scale)
        out = np.where(np.abs(distance) < 1.0, distance / F32(1.0), np.sign(distance) * (F32(2.0) - t))
        return ((out / F32(2.0)) * F32(2.0)).astype(F32)   # Identity.inverse with default factors

    def contract_points(self, points):                 # contract.py:178-192
        points = points / F32(self.r0)
        dist = np.sqrt(np.sum(points * points, axis=-1, keepdims=True, dtype=F32))
        with np.errstate(divide='ignore', invalid='ignore'):
            inv = F32(1.0) / np.abs(dist)
            inv_end = self.r0 / self.r1
            scale = 1.0 / (1.0 - inv_end)
            t = (inv - F32(inv_end)) * F32(scale)
            outer = (points / dist) * (F32(2.0) - t)
        return np.where(dist < 1, points, outer).astype(F32)

    def contract_points_and_distance(self, rays_o, points, distance):   # contract.py:43-50
        o_c = self.contract_points(rays_o)
        p_c = self.contract_points(points)
        diff = p_c - o_c[..., None, :]
        distance = np.sqrt(np.sum(diff * diff, axis=-1, keepdims=True, dtype=F32))
        return p_c, distance


class IdentityContract:
    """nlf/contract.py:53-62 (and the default when the YAML has no `contract`)."""
    contract_samples = False

    def __init__(self, cfg=None, dataset=None):
        self.contract_samples = bool((cfg or {}).get('contract_samples', False))

    def inverse_contract_distance(self, d):
        return d

    def contract_distance(self, d):
        return _f(d)

    def contract_points_and_distance(self, rays_o, points, distance):
        return points, distance


class _AffineContract:
    """Shared shape of BBoxContract / ZDepthContract: BaseContract.contract_points_and_distance
    (contract.py:43-50) contracts the ray origin and the points and re-measures the distance."""
    contract_samples = False

    def contract_points_and_distance(self, rays_o, points, distance):
        o_c = self.contract_points(rays_o)
        p_c = self.contract_points(points)
        diff = p_c - o_c[..., None, :]
        return p_c, np.sqrt(np.sum(diff * diff, axis=-1, keepdims=True, dtype=F32))


class BBoxContract(_AffineContract):
    """nlf/contract.py:65-87 (the dataset is never read, whatever use_dataset_bounds says)."""

    def __init__(self, cfg, dataset=None):
        self.contract_samples = bool(cfg.get('contract_samples', False))
        self.bbox_min = _f(cfg.get('bbox_min', [-1.0, -1.0, -1.0]))
        self.bbox_max = _f(cfg.get('bbox_max', [1.0, 1.0, 1.0]))
        self.fac = F32(np.mean(np.abs(self.bbox_max - self.bbox_min), dtype=F32))

    def inverse_contract_distance(self, d):
        return (d * self.fac).astype(F32)

    def contract_distance(self, d):
        return (_f(d) / self.fac).astype(F32)

    def contract_points(self, p):
        return ((p - self.bbox_min) / (self.bbox_max - self.bbox_min)).astype(F32)


class ZDepthContract(_AffineContract):
    """nlf/contract.py:90-111."""

    def __init__(self, cfg, dataset):
        self.contract_samples = bool(cfg.get('contract_samples', False))
        if cfg.get('use_dataset_bounds', False):
            r1 = cfg.get('contract_end_radius', dataset['depth_range'][1])
        else:
            r1 = cfg.get('contract_end_radius', float('inf'))
        self.fac = r1 / 2.0                               # python float

    def inverse_contract_distance(self, d):
        return (d * F32(self.fac)).astype(F32)

    def contract_distance(self, d):
        return (_f(d) / F32(self.fac)).astype(F32)

    def contract_points(self, p):
        return (p / F32(self.fac)).astype(F32)


def _torch_pow(x, exponent):
    """torch.pow(float32 tensor, python scalar) on CPU: exponents 0.5 / 2 / 3 / -0.5 / -1 / -2 are sqrt / products / reciprocals
    (ATen pow_tensor_scalar_optimized_kernel), the rest float32 powf."""
    x = np.asarray(x, F32)
    e = float(exponent)
    with np.errstate(divide='ignore', invalid='ignore'):
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


class DoNeRFContract:
    """nlf/contract.py:195-240."""

    def __init__(self, cfg, dataset):
        self.contract_samples = bool(cfg.get('contract_samples', False))
        if cfg.get('use_dataset_bounds', False):
            r0 = cfg.get('contract_start_radius', max(dataset['depth_range'][0] * 1.75, 1.0))
            r1 = cfg.get('contract_end_radius', dataset['depth_range'][1] * 1.5)
        else:
            r0 = cfg.get('contract_start_radius', None)
            r1 = cfg.get('contract_end_radius', 10000.0)
        if r0 is None:
            self.power = float(cfg.get('power', 2.0))
            self.fac = float(np.power(2.0, self.power) / r1)
        else:
            self.fac = 1.0 / r0
            self.power = float(np.log(r1 / r0) / np.log(2.0))
        if 'distance_activation' in cfg:
            raise NotImplementedError('distance_activation is outside the hot-path scope')

    def inverse_contract_distance(self, distance):     # contract.py:226-230
        distance = (_f(distance) / F32(2.0)) * F32(2.0)    # identity activation
        distance = np.clip(distance, F32(-2.0), F32(2.0))
        return (_torch_pow(np.abs(distance) + F32(1e-8), self.power) * np.sign(distance) / F32(self.fac)).astype(F32)

    def contract_distance(self, distance):             # contract.py:232-236
        distance = _f(distance) * F32(self.fac)
        distance = _torch_pow(np.abs(distance) + F32(1e-8), 1.0 / self.power) * np.sign(distance)
        return ((distance / F32(2.0)) * F32(2.0)).astype(F32)

    def contract_points(self, points):                 # contract.py:238-240
        dists = np.sqrt(np.sum(points * points, axis=-1, keepdims=True, dtype=F32))
        with np.errstate(divide='ignore', invalid='ignore'):
            return ((points / dists) * _torch_pow(dists * F32(self.fac) + F32(1e-8), 1.0 / self.power)).astype(F32)

    def contract_points_and_distance(self, rays_o, points, distance):   # BaseContract, contract.py:43-50
        o_c = self.contract_points(rays_o)
        p_c = self.contract_points(points)
        diff = p_c - o_c[..., None, :]
        return p_c, np.sqrt(np.sum(diff * diff, axis=-1, keepdims=True, dtype=F32))


def make_contract(cfg, dataset):
    if cfg is None:
        return IdentityContract()
    t = cfg['type']
    if 'stop_iters' in cfg:
        raise NotImplementedError('contract.stop_iters')
    if t == 'mipnerf':
        return MipNerfContract(cfg, dataset)
    if t == 'identity':
        return IdentityContract(cfg, dataset)
    if t == 'bbox':
        return BBoxContract(cfg, dataset)
    if t == 'z_depth':
        return ZDepthContract(cfg, dataset)
    if t == 'donerf':
        return DoNeRFContract(cfg, dataset)
    raise NotImplementedError(f'contract {t} is outside the hot-path scope')


# --------------------------------------------------------------------------- closed-form intersections
def _safe_dir(d):
    """utils/intersect_utils.py:136-140: |d| < 1e-5 -> 1e12."""
    return np.where(np.abs(d) < F32(1e-5), F32(1e12), d).astype(F32)


def intersect_axis_plane(rays, val, dim):              # intersect_utils.py:127-150
    o, d = rays[..., :3], _safe_dir(rays[..., 3:6])
    return ((val - o[..., dim]) / d[..., dim]).astype(F32)


def _quadratic(o, d, radius):                          # intersect_utils.py:45-84 / 86-125
    oo = np.sum(o * o, -1, dtype=F32)
    dd = np.sum(d * d, -1, dtype=F32)
    od = np.sum(o * d, -1, dtype=F32)
    a = dd
    b = F32(2) * od
    c = oo - radius * radius
    disc = b * b - F32(4) * a * c
    disc = np.where(disc < 0, F32(0), disc)
    sq = np.sqrt(disc + F32(1e-8))
    with np.errstate(divide='ignore', invalid='ignore'):
        t1 = (-b + sq) / (F32(2) * a)
        t2 = (-b - sq) / (F32(2) * a)
    t1 = np.where(disc <= 0, F32(0), t1)
    t2 = np.where(disc <= 0, F32(0), t2)
    return np.where((t2 < 0) | (radius < 0), t1, t2).astype(F32)


def intersect_sphere(rays, radius):
    return _quadratic(rays[..., 0:3], rays[..., 3:6], radius)


def intersect_cylinder(rays, radius):
    o = np.stack([rays[..., 0], rays[..., 2]], -1)
    d = np.stack([rays[..., 3], rays[..., 5]], -1)
    return _quadratic(o, d, radius)


def _normalize(v):
    """F.normalize(p=2, dim=-1, eps=1e-12)."""
    n = np.sqrt(np.sum(v * v, -1, keepdims=True, dtype=F32))
    return (v / np.maximum(n, F32(1e-12))).astype(F32)


def _norm(v):
    return np.sqrt(np.sum(v * v, -1, dtype=F32))


def pluecker_pos(o, d):                                # nlf/param.py:297-307: closest point of the line to 0
    d = _normalize(d)
    m = np.cross(o, d).astype(F32)
    return np.cross(d, m).astype(F32)


def _xz(v):                                            # [x, 0, z]
    return np.stack([v[..., 0], np.zeros_like(v[..., 1]), v[..., 2]], -1)


def pluecker_pos_cylinder(o, d):                       # nlf/param.py:310-322
    return pluecker_pos(_xz(o), _xz(d))


def _signed_base_distance(d, diff):                    # sign(d . diff) * |diff|
    return (np.sign(np.sum(d * diff, -1, dtype=F32)) * _norm(diff)).astype(F32)


# --------------------------------------------------------------------------- grid_sample
def grid_sample_2d(plane, gx, gy):
    """F.grid_sample(plane[None], grid, mode='bilinear', padding_mode='zeros',
    align_corners=True) for plane (C,H,W) and N sample points -> (C,N)."""
    C, H, W = plane.shape
    ix = ((gx + F32(1)) / F32(2)) * F32(W - 1)
    iy = ((gy + F32(1)) / F32(2)) * F32(H - 1)
    x0f = np.floor(ix)
    y0f = np.floor(iy)
    x0 = x0f.astype(np.int64)
    y0 = y0f.astype(np.int64)
    x1 = x0 + 1
    y1 = y0 + 1
    # ATen weights: nw=(x1-ix)(y1-iy) ne=(ix-x0)(y1-iy) sw=(x1-ix)(iy-y0) se=(ix-x0)(iy-y0)
    x1f = x0f + F32(1)
    y1f = y0f + F32(1)
    wnw = (x1f - ix) * (y1f - iy)
    wne = (ix - x0f) * (y1f - iy)
    wsw = (x1f - ix) * (iy - y0f)
    wse = (ix - x0f) * (iy - y0f)
    out = np.zeros((C, gx.shape[0]), F32)

    def acc(xi, yi, w):
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        xc = np.clip(xi, 0, W - 1)
        yc = np.clip(yi, 0, H - 1)
        v = plane[:, yc, xc]
        return np.where(ok[None], v * w[None], F32(0)).astype(F32)

    out = out + acc(x0, y0, wnw)
    out = out + acc(x1, y0, wne)
    out = out + acc(x0, y1, wsw)
    out = out + acc(x1, y1, wse)
    return out


# --------------------------------------------------------------------------- SH
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
      -1.0925484305920792, 0.5462742152960396]


def eval_sh_bases_deg2(dirs):                          # utils/sh_utils.py:94-119
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    out = np.empty(dirs.shape[:-1] + (9,), F32)
    out[..., 0] = C0
    out[..., 1] = F32(-C1) * y
    out[..., 2] = F32(C1) * z
    out[..., 3] = F32(-C1) * x
    xx, yy, zz = x * x, y * y, z * z
    xy, yz, xz = x * y, y * z, x * z
    out[..., 4] = F32(C2[0]) * xy
    out[..., 5] = F32(C2[1]) * yz
    out[..., 6] = F32(C2[2]) * (F32(2.0) * zz - xx - yy)
    out[..., 7] = F32(C2[3]) * xz
    out[..., 8] = F32(C2[4]) * (xx - yy)
    return out


# --------------------------------------------------------------------------- ray_intersect
class _Isect:
    """One `ray_intersect` stage (Intersect.__init__ + forward, nlf/intersect/base.py:52-259 and the
    subclasses in z.py / primitive.py / voxel.py).  A model has one; point_prediction cascades two."""

    def __init__(self, ecfg, Z_expected, ds):
        self.Z, self.ds = Z_expected, ds
        c = ecfg['intersect']
        self.isect = c
        self.isect_type = c['type']
        Z = int(ecfg['z_channels'])
        assert Z == self.Z
        ds = self.ds
        udb = c.get('use_dataset_bounds', False)
        self.near = c['near'] if 'near' in c else (ds['near'] if udb else 0.0)      # base.py:88-94
        self.far = c.get('far', float('inf'))
        self.origin = _f(c.get('origin', [0.0, 0.0, 0.0]))
        self.contract = make_contract(c.get('contract'), ds)
        self.z_act = Act(c.get('activation'))
        self.use_sigma = c.get('use_sigma', False)
        self.in_density_field = c.get('in_density_field', 'sigma')
        self.sort = c.get('sort', False)
        for k in ('weight_fn', 'sort_outputs', 'dropout', 'normalize', 'residual_z',
                  'residual_distance', 'clamp', 'use_local_prediction', 'flip_axes', 'max_axis'):
            if c.get(k):
                raise NotImplementedError(f'intersect option {k} is outside the hot-path scope')
        if c.get('num_repeat', 1) != 1:
            raise NotImplementedError('intersect option num_repeat')
        if c.get('use_disparity', False):
            raise NotImplementedError('use_disparity')
        # base.py:104-108,197-198: the near/far mask is dropped once cur_iter > mask.stop_iters (inference: 1e7)
        stop = c['mask'].get('stop_iters', float('inf')) if 'mask' in c else float('inf')
        self.mask_on = not ((10_000_000 if ITERATION is None else ITERATION) > stop)
        t = self.isect_type
        if t == 'voxel_grid':                           # voxel.py:19-70: Z/3 planes per axis
            self._setup_voxel_grid(c, udb)
            return
        if t == 'deformable_voxel_grid':                # voxel.py:115-176
            if udb:
                raise NotImplementedError('deformable_voxel_grid with use_dataset_bounds')
            self.dvg_normals = _f(c.get('start_normal', [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]))
            na = self.dvg_normals.shape[0]
            self.dvg_scale = F32(c.get('normal_scale_factor', 0.1))
            initial, end = _f(c.get('initial', [0.0, 0.0, 0.0])), _f(c.get('end', [1.0, 1.0, 1.0]))
            if self.contract.contract_samples:
                initial, end = self.contract.contract_distance(initial), self.contract.contract_distance(end)
            nz = Z // na
            self.samples = np.stack([torch_linspace(initial[d], end[d], nz) for d in range(na)], -1).reshape(-1)
            if 'z_scale' in c:
                zs = _f(c['z_scale'])[0]
            else:
                zs = np.abs(self.samples[1] - self.samples[0]) if nz > 1 else F32(1.0)   # on the flattened (Z,1) tensor
            self.z_scale = F32(1.0) if zs == 0 else F32(zs)
            return
        if t == 'z_plane':                              # z.py:25-71
            if udb:
                initial, end = F32(-ds['near']), F32(-ds['far'])
            else:
                initial, end = F32(c.get('initial', 0.0)), F32(c.get('end', 1.0))
        elif t in ('sphere', 'cylinder'):               # primitive.py:185-215, 370-400
            if udb:
                initial = F32(c['initial']) if 'initial' in c else F32(ds['near'] * 1.5)
                end = F32(c['end']) if 'end' in c else F32(ds['far'] * 1.5)
            else:
                initial, end = F32(c.get('initial', 0.0)), F32(c.get('end', 1.0))
            self.origin_scale = F32(c.get('origin_scale_factor', 0.0))
            self.origin_initial = _f(c.get('origin_initial', [1.0, 1.0, 1.0]))
        elif t in ('sphere_new', 'cylinder_new'):       # primitive.py:256-303, 441-488
            if udb:
                if c['outward_facing']:
                    initial = F32(c['initial']) if 'initial' in c else F32(ds['near'] * 1.5)
                else:
                    initial = F32(c['initial']) if 'initial' in c else F32(-ds['far'] * 1.5)
                end = F32(c['end']) if 'end' in c else F32(ds['far'] * 1.5)
            else:
                initial, end = F32(c.get('initial', 0.0)), F32(c.get('end', 1.0))
            self.origin_scale = F32(c.get('origin_scale_factor', 0.0))
            self.resize_scale = F32(c.get('resize_scale_factor', 0.0))
            self.resize_initial = _f(c.get('resize_initial', [1.0, 1.0, 1.0]))
        elif t == 'euclidean_distance_unified':         # primitive.py:131-160
            if udb:
                initial = F32(c['initial']) if 'initial' in c else F32(-ds['far'])
                end = F32(c['end']) if 'end' in c else F32(ds['far'])
            else:
                initial, end = F32(c.get('initial', 0.0)), F32(c.get('end', 1.0))
        else:
            raise NotImplementedError(f'intersect {t} is outside the hot-path scope (SURVEY 8f-1; `plane` and '
                                      f'`euclidean_distance` cannot run in the reference: scalar z_scale, base.py:129)')
        if self.contract.contract_samples:
            initial = self.contract.contract_distance(initial)
            end = self.contract.contract_distance(end)
        self.samples = torch_linspace(initial, end, Z)
        if Z > 1:
            if 'z_scale' in c:
                self.z_scale = F32(c['z_scale'])
            elif 'num_samples_for_scale' in c and t == 'z_plane':        # z.py:63-65
                self.z_scale = np.abs(self.samples[1] - self.samples[0]) * F32(Z / float(c['num_samples_for_scale']))
            else:
                self.z_scale = np.abs(self.samples[1] - self.samples[0])
        else:
            self.z_scale = F32(c.get('z_scale', 1.0))

    def _setup_voxel_grid(self, c, udb):
        if self.Z % 3:
            raise ValueError('voxel_grid needs z_channels divisible by 3')
        nz = self.Z // 3
        fac = c.get('fac', 1.0)
        if udb:                                         # voxel.py:27-29: dataset bbox * fac
            initial = _f(c['initial']) if 'initial' in c else (_f(self.ds['bbox_min']) * F32(fac)).astype(F32)
            end = _f(c['end']) if 'end' in c else (_f(self.ds['bbox_max']) * F32(fac)).astype(F32)
        else:
            initial, end = _f(c.get('initial', [0.0, 0.0, 0.0])), _f(c.get('end', [1.0, 1.0, 1.0]))
        if self.contract.contract_samples:
            initial = self.contract.contract_distance(initial)
            end = self.contract.contract_distance(end)
        self.voxel_samples = np.stack([torch_linspace(initial[d], end[d], nz) for d in range(3)], -1)   # (Z/3, 3)
        if 'z_scale' in c:
            zs = _f(c['z_scale'])
        elif nz > 1:
            zs = np.abs(self.voxel_samples[1] - self.voxel_samples[0])
        else:
            zs = np.ones(3, F32)
        self.voxel_scale = np.where(zs == 0, F32(1.0), zs).astype(F32)
        self.outward_facing = bool(c.get('outward_facing', False))

    def _process_scalar_z(self, z):                     # base.py:128-140
        z = z * self.z_scale + self.samples[None]
        if self.contract.contract_samples:
            z = self.contract.inverse_contract_distance(z)
        return z.astype(F32)

    def __call__(self, rays, x):                        # base.py:142-259
        B = rays.shape[0]
        r = np.concatenate([rays[:, :3] - self.origin[None], rays[:, 3:6]], -1).astype(F32)
        zv = x['z_vals']                                # (B,Z,zc) already through its head activation
        if self.use_sigma and self.in_density_field in x:
            sigma = x[self.in_density_field].reshape(B, -1)
        else:
            sigma = np.zeros((B, self.Z), F32)
        zv = self.z_act(zv.reshape(B, self.Z, -1)) * (F32(1) - sigma[..., None])
        t = self.isect_type
        if t == 'z_plane':
            z = self._process_scalar_z(zv.reshape(B, self.Z))
            dists = intersect_axis_plane(r[:, None, :], z, 2)                 # z.py:88-95
        elif t in ('sphere', 'cylinder'):
            origins = zv[..., :3] * self.origin_scale + self.origin_initial[None, None]
            radii = self._process_scalar_z(zv[..., 3])
            rr = np.concatenate([r[:, None, 0:3] * origins, r[:, None, 3:6] * origins], -1)
            dists = intersect_sphere(rr, radii) if t == 'sphere' else intersect_cylinder(rr, radii)
        elif t in ('sphere_new', 'cylinder_new'):       # primitive.py:305-363, 490-545
            origins = zv[..., :3] * self.origin_scale
            resize = zv[..., 3:6] * self.resize_scale + self.resize_initial[None, None]
            raw_offsets = self._process_scalar_z(zv[..., 6])
            radii = self._process_scalar_z(zv[..., 7])
            ro = ((r[:, None, 0:3] - origins) * resize).astype(F32)
            rd = (r[:, None, 3:6] * resize).astype(F32)
            rn = _normalize(rd)
            rr = np.concatenate([ro, rn], -1)
            if t == 'sphere_new':
                tt = intersect_sphere(rr, radii)
                base_pos = pluecker_pos(ro, rn)
                min_radius = _norm(base_pos)             # min_sphere_radius, intersect_utils.py:27-33
                base_distance = _signed_base_distance(rn, base_pos - ro)
            else:
                tt = intersect_cylinder(rr, radii)
                base_pos = pluecker_pos_cylinder(ro, rn)
                min_radius = _norm(base_pos)             # min_cylinder_radius, intersect_utils.py:35-43
                o_c, d_c = _xz(ro), _xz(rn)
                with np.errstate(divide='ignore', invalid='ignore'):
                    base_distance = (_signed_base_distance(d_c, base_pos - o_c) / _norm(d_c)).astype(F32)
            recycle = np.abs(radii) < min_radius + F32(4) * self.z_scale       # samples on missed primitives
            tt = np.where(recycle, raw_offsets + base_distance, tt)
            dists = (tt / (_norm(rd) + F32(1e-5))).astype(F32)
        elif t == 'euclidean_distance_unified':         # primitive.py:162-176
            z = self._process_scalar_z(zv.reshape(B, self.Z))
            diff = pluecker_pos(r[:, :3], r[:, 3:6]) - r[:, :3]
            dists = (z + _signed_base_distance(r[:, 3:6], diff)[:, None]).astype(F32)
        elif t == 'deformable_voxel_grid':              # voxel.py:178-213, intersect_utils.py:210-236
            na = self.dvg_normals.shape[0]
            d = self._process_scalar_z(zv[..., 3])
            normal = zv[..., :3].reshape(-1, na, 3) * self.dvg_scale + self.dvg_normals[None]
            normal = _normalize(normal.reshape(B, -1, 3))
            o_n = np.sum(r[:, None, :3] * normal, -1, dtype=F32)
            d_n = np.sum(r[:, None, 3:6] * normal, -1, dtype=F32)
            d_n = np.where(np.abs(d_n) < F32(1e-5), F32(1e12), d_n).astype(F32)
            dists = ((d - o_n) / d_n).astype(F32)
        elif t == 'voxel_grid':                         # voxel.py:72-112, intersect_utils.py:152-179
            nz = self.Z // 3
            z = zv.reshape(B, nz, 3) * self.voxel_scale[None, None] + self.voxel_samples[None]     # base.py:129
            z = z.reshape(B, -1)
            if self.contract.contract_samples:
                z = self.contract.inverse_contract_distance(z)
            z = z.astype(F32).reshape(B, nz, 3)
            if self.outward_facing:
                z = z * np.sign(r[:, None, 3:6])
            dists = ((z - r[:, None, 0:3]) / _safe_dir(r[:, None, 3:6])).astype(F32).reshape(B, self.Z)
        else:
            raise NotImplementedError(t)
        if self.mask_on:
            mask = (dists <= F32(self.near)) | (dists >= F32(self.far))         # base.py:194
            dists = np.where(mask, F32(0), dists)
        if self.sort:
            dists = np.sort(dists, axis=1)                                    # only dists is permuted
        dists = dists[..., None]
        mask = dists == 0
        points = (r[:, None, :3] + r[:, None, 3:6] * dists).astype(F32)
        x['raw_points'], x['raw_distance'] = points, dists
        points, dists = self.contract.contract_points_and_distance(r[:, :3], points, dists)
        dists = np.where(mask, F32(0), dists).astype(F32)
        x['points'] = points.astype(F32)
        x['distances'] = dists
        x['weights'] = np.ones_like(dists)
        return x


# --------------------------------------------------------------------------- the model
class HyperReelOracle:
    """cfg: the `experiment.model` YAML as a plain dict, AFTER the *_epoch->*_iter
    rewrite is irrelevant here (inference).  dataset: {near, far, depth_range,
    num_keyframes, num_frames}.  sd: {reference state_dict key: ndarray}, keys as
    listed in SURVEY.md section 5 without the leading `render_fn.`."""

    EMB = 'model.embedding_model.embeddings.'
    NET = 'model.color_model.net.'

    def __init__(self, cfg, dataset, sd, iteration=None):
        """iteration: training iteration of the EaseValue / WindowedPE schedules (what INRSystem.set_train_iter hands to
        set_iter, nlf/__init__.py:608-614); None = converged, the state render / test run in (:582-583).  `cfg` must
        carry the *_iters keys (hyperreel_amd.config.epoch_to_iter / nlf/__init__.py:305-315) for it to matter."""
        global ITERATION
        ITERATION = self.iteration = iteration
        self.cfg = cfg
        self.ds = dataset
        self.sd = {k: _f(v) for k, v in sd.items() if np.asarray(v).dtype.kind == 'f'}
        if cfg.get('param', {}).get('fn', 'identity') != 'identity':
            raise NotImplementedError('model-level ray param other than identity')
        self.stages = []
        for idx, (key, ecfg) in enumerate(cfg['embedding']['embeddings'].items()):
            self.stages.append((idx, ecfg['type'], ecfg))
        self._setup_prediction()
        # one _Isect per ray_intersect stage; the sample count of a stage is that of the prediction feeding it
        self._isects = {}
        Z = self.Z
        for idx, typ, ecfg in self.stages:
            if typ == 'point_prediction':
                Z = int(ecfg.get('out_z_channels', 1))
            elif typ == 'ray_intersect':
                self._isects[idx] = _Isect(ecfg, Z, self.ds)
        self._setup_point_prediction()
        last = self._isects[max(self._isects)]
        for k, v in vars(last).items():                  # single-stage view kept for tests / tools: orc.samples, orc.near ...
            if k not in ('Z', 'ds'):
                setattr(self, k, v)
        self.Z_final = last.Z
        self._setup_color()

    # ---- ray_prediction -------------------------------------------------------------
    def _setup_prediction(self):
        (idx, _, ecfg), = [s for s in self.stages if s[1] == 'ray_prediction']
        self.pred_idx = idx
        self.pred_cfg = ecfg
        self.Z = int(ecfg['z_channels'])
        self.out_names = list(ecfg['outputs'].keys())
        self.out_shapes = [int(ecfg['outputs'][k]['channels']) for k in self.out_names]
        self.out_acts = [Act(ecfg['outputs'][k].get('activation')) for k in self.out_names]
        if ecfg.get('ray_outputs'):
            raise NotImplementedError('ray_outputs are outside the hot-path scope')
        net = ecfg['net']
        self.zero_net = net['type'] == 'zero'          # ZeroMLP, nlf/nets/mlp.py:14-33
        if self.zero_net:
            self.D, self.skips, self.layers = 0, [], []
            return
        if net['type'] != 'base':
            raise NotImplementedError(f"net {net['type']} is outside the hot-path scope")
        self.D, self.skips, self.layers = self._load_layers(idx, net)

    def _load_layers(self, idx, net):
        D = int(net['depth']) - 2                      # ray.py:283-285 / point.py:117-119
        layers = []
        pre = f'{self.EMB}{idx}.net.layers.'
        for i in range(D + 2):
            mid = '.0' if i < D + 1 else ''            # Sequential(Linear, act) vs bare Linear
            layers.append((self.sd[f'{pre}{i}{mid}.weight'], self.sd[f'{pre}{i}{mid}.bias']))
        return D, list(net.get('skips', [])), layers

    def _param_pe(self, rays):
        return self._param_pe_cfg(self.pred_cfg['params'], rays)

    def _param_pe_cfg(self, params, rays):
        cols = []
        for pkey, pcfg in params.items():
            x = rays[:, pcfg['start']:pcfg['end']]
            p = pcfg['param']
            fn = p['fn']
            if fn == 'identity':
                y = x
            elif fn == 'pluecker':                     # param.py:244-253
                origin = _f(p.get('origin', [0.0, 0.0, 0.0]))
                o = x[:, :3] - origin[None]
                d = x[:, 3:6]
                n = np.sqrt(np.sum(d * d, -1, keepdims=True, dtype=F32))
                d = d / np.maximum(n, F32(1e-12))
                if p.get('use_local_param', False):
                    raise NotImplementedError
                m = np.cross(o, d).astype(F32)
                y = np.concatenate([d * F32(p.get('direction_multiplier', 1.0)),
                                    m * F32(p.get('moment_multiplier', 1.0))], -1)
            elif fn == 'two_plane':                    # param.py:87-115
                origin = _f(p.get('origin', [0.0, 0.0, 0.0]))
                r = np.concatenate([x[:, :3] - origin[None], x[:, 3:6]], -1)
                if p.get('use_local_param', False):
                    raise NotImplementedError
                t1 = intersect_axis_plane(r, F32(p.get('near', -1.0)), 2)
                t2 = intersect_axis_plane(r, F32(p.get('far', 0.0)), 2)
                y = np.concatenate([r[:, :2] + r[:, 3:5] * t1[:, None],
                                    r[:, :2] + r[:, 3:5] * t2[:, None]], -1)
            else:
                raise NotImplementedError(f'ray param {fn} is outside the hot-path scope')
            y = y.astype(F32)
            pe = pcfg.get('pe')
            if pe is not None:                         # pe.py:210-221 / 53-66
                if pe['type'] not in ('windowed', 'basic'):
                    raise NotImplementedError(pe['type'])
                n = int(pe['n_freqs'])
                fm = pe.get('freq_multiplier', 2.0)
                bm = F32(pe.get('base_multiplier', 1.0)) if pe['type'] == 'windowed' else F32(1.0)
                freqs = (F32(fm) ** torch_linspace(1.0, float(n), n)).astype(F32)
                out = [] if pe.get('exclude_identity', False) and pe['type'] == 'windowed' else [y]
                if pe['type'] == 'windowed':
                    wts = windowed_pe_weights(pe, ITERATION)
                    for f, w in zip(freqs, wts):
                        if w == 1.0:
                            out += [np.sin(bm * f * y), np.cos(bm * f * y)]
                        else:                              # pe.py:216-218
                            out += [F32(w) * np.sin(bm * f * y), F32(w) * np.cos(bm * f * y)]
                elif n > 0:                            # BasicPE: [x, sin(all), cos(all)], channel-major
                    cur = (freqs[None, None] * y[..., None]).reshape(y.shape[0], -1)
                    out += [np.sin(cur), np.cos(cur)]
                y = np.concatenate(out, -1).astype(F32)
            cols.append(y)
        return np.concatenate(cols, -1).astype(F32)

    def mlp(self, x):                                  # mlp.py:159-172
        return self._run_mlp(x, self.D, self.skips, self.layers)

    @staticmethod
    def _run_mlp(x, D, skips, layers):
        inp = x
        for i, (w, b) in enumerate(layers):
            if i in skips:
                x = np.concatenate([inp, x], -1)
            x = (x @ w.T + b).astype(F32)
            if i < D + 1:
                x = np.where(x >= 0, x, x * F32(0.01)).astype(F32)
        return x

    # ---- point_prediction (cascades) ---------------------------------------------------
    def _setup_point_prediction(self):                 # point.py:39-135
        self._pp = {}
        for idx, typ, e in self.stages:
            if typ != 'point_prediction':
                continue
            if e.get('filter', False):
                raise NotImplementedError('point_prediction.filter')
            if e.get('rays_name', 'rays') != 'rays' or e.get('points_name', 'points') != 'points':
                raise NotImplementedError('point_prediction with renamed rays / points')
            if e['net']['type'] != 'base':
                raise NotImplementedError(f"point_prediction net {e['net']['type']}")
            outs = e['outputs']
            if any(o.get('residual', False) for o in outs.values()):
                raise NotImplementedError('residual point_prediction outputs')
            D, skips, layers = self._load_layers(idx, e['net'])
            self._pp[idx] = {
                'cfg': e, 'D': D, 'skips': skips, 'layers': layers, 'names': list(outs.keys()),
                'shapes': [int(o['channels']) for o in outs.values()], 'acts': [Act(o.get('activation')) for o in outs.values()]}

    def _point_prediction(self, idx, rays, x):         # point.py:137-203
        pp = self._pp[idx]
        e = pp['cfg']
        pts = x['points']
        B, Zi = pts.shape[:2]
        cols = []
        for name, n in e['inputs'].items():
            if name == 'viewdirs':
                cols.append(np.repeat(rays[:, None, 3:6], Zi, 1))
            elif name == 'origins':
                cols.append(np.repeat(rays[:, None, 0:3], Zi, 1))
            elif name == 'times':
                cols.append(np.repeat(rays[:, None, -1:], Zi, 1))
            else:
                cols.append(x[name][..., :int(n)])
        inp = np.concatenate(cols, -1).astype(F32).reshape(B * Zi, -1)
        h = self._run_mlp(self._param_pe_cfg(e['params'], inp), pp['D'], pp['skips'], pp['layers'])
        x['_head_raw_points'] = h
        h = h.reshape(B, -1, sum(pp['shapes']))
        o = 0
        for name, n, act in zip(pp['names'], pp['shapes'], pp['acts']):
            x[name] = act(h[..., o:o + n])
            o += n
        return x

    def _predict(self, rays, x):                       # ray.py:316-347
        if self.zero_net:
            h = np.zeros((rays.shape[0], self.Z * sum(self.out_shapes)), F32)
        else:
            h = self.mlp(self._param_pe(rays))
        x['_head_raw'] = h
        h = h.reshape(rays.shape[0], self.Z, -1)
        o = 0
        for name, n, act in zip(self.out_names, self.out_shapes, self.out_acts):
            x[name] = act(h[..., o:o + n])
            o += n
        return x

    # ---- point stages ------------------------------------------------------------------
    def _advect(self, rays, x, ecfg):                   # point.py:780-831, flow_utils.py:10-35
        if ecfg.get('use_angular_flow', False):
            raise NotImplementedError('angular flow')
        t = rays[:, -1:]
        K, Fr = self.ds['num_keyframes'], self.ds['num_frames']
        if K > 0:
            fac = K * (Fr - 1) / Fr
            tt = t * F32(fac)
            base_t = (np.round(np.clip(tt, F32(0.0), F32(K - 1.0)) - F32(1e-5)) * F32(1.0 / fac)).astype(F32)
        else:
            base_t = np.zeros_like(t)
        toff = (t - base_t)[:, None, :]
        points = x['points']
        if ecfg.get('use_spatial_flow', False):
            flow = Act(ecfg.get('spatial_flow_activation'))(x['spatial_flow'])
            x['spatial_flow'] = flow
            points = points + flow * toff
        x['points'] = points.astype(F32)
        Zc = points.shape[1]
        x['base_times'] = np.repeat(base_t[:, None, :], Zc, 1)
        x['time_offset'] = np.repeat(toff, Zc, 1)
        return x

    def _point_offset(self, x, ecfg):                    # point.py:371-396
        field = ecfg.get('in_density_field', 'sigma')
        if ecfg.get('use_sigma', True) and field in x:
            sigma = x[field]
        else:
            sigma = np.zeros(x['points'].shape[:2] + (1,), F32)
        off = Act(ecfg.get('activation'))(x[ecfg.get('in_offset_field', 'point_offset')]) * (F32(1) - sigma)
        x['points'] = (x['points'] + off).astype(F32)
        return x

    def embed(self, rays):
        global ITERATION
        ITERATION = self.iteration                        # stages build their activations while they run
        """LightfieldModel.embed (models.py:131-133), un-flattened (B,Z,k) fields."""
        rays = _f(rays)
        x = {'rays': rays}
        for idx, typ, ecfg in self.stages:
            if typ == 'ray_prediction':
                x = self._predict(rays, x)
            elif typ == 'ray_intersect':
                x = self._isects[idx](rays, x)
            elif typ == 'point_prediction':
                x = self._point_prediction(idx, rays, x)
            elif typ == 'advect_points':
                x = self._advect(rays, x, ecfg)
            elif typ == 'point_offset':
                x = self._point_offset(x, ecfg)
            elif typ == 'add_point_outputs':             # point.py:857-869
                eo = ecfg['extra_outputs']
                Zc = x['points'].shape[1]
                if 'times' in eo and 'times' not in x:
                    x['times'] = np.repeat(rays[:, None, -1:], Zc, 1)
                if 'base_times' in eo and 'base_times' not in x:
                    x['base_times'] = np.repeat(rays[:, None, -1:], Zc, 1)
                if 'viewdirs' in eo and 'viewdirs' not in x:
                    x['viewdirs'] = np.repeat(rays[:, None, 3:6], Zc, 1)
            elif typ == 'color_transform':              # point.py:585-596: a no-op unless dataset.val_all
                if self.ds.get('val_all', False):
                    tab = self.sd[f'{self.EMB}{idx}.color_embedding']
                    ids = np.round(rays[:, -2]).astype(np.int64)
                    row = tab[ids]
                    x[ecfg.get('out_transform_field', 'color_transform_global')] = Act(ecfg.get('transform_activation'))(row[:, :9])
                    x[ecfg.get('out_shift_field', 'color_shift_global')] = Act(ecfg.get('shift_activation'))(row[:, -3:])
            elif typ == 'extract_fields':               # point.py:236-244: the colour net only sees these
                x['_extracted'] = set(ecfg['fields'])
            else:
                raise NotImplementedError(f'embedding {typ} is outside the hot-path scope')
        return x

    # ---- colour -----------------------------------------------------------------------
    def _setup_color(self):
        n = self.cfg['color']['net']
        self.net_type = n['type']
        self.video = self.net_type == 'tensor_vm_split_time'
        if self.net_type not in ('tensor_vm_split_no_sample', 'tensor_vm_split_time'):
            raise NotImplementedError(self.net_type)
        aabb = self.sd.get(self.NET + 'aabb')
        self.aabb = _f(n['aabb']) if aabb is None else aabb
        self.inv_size = (F32(2.0) / (self.aabb[1] - self.aabb[0])).astype(F32)
        self.distance_scale = F32(n.get('distance_scale', 25))
        self.thr = F32(n.get('rm_weight_mask_thre', 0.0001))
        self.act = n.get('fea2denseAct', 'softplus')
        self.density_shift = F32(n.get('density_shift', -10.0))
        self.shading = n.get('shadingMode', 'MLP_PE')
        if self.shading not in ('RGB', 'SH'):
            raise NotImplementedError(self.shading)
        self.white_bg = bool(n.get('white_bg', 0)) and not bool(n.get('black_bg', 0))
        if self.video and n.get('densityMode', 'Density') != 'Density':
            raise NotImplementedError(n.get('densityMode'))
        g = lambda k: self.sd[self.NET + k]
        if self.video:
            self.d_space = [g(f'density_plane_space.{i}')[0] for i in range(3)]
            self.d_time = [g(f'density_plane_time.{i}')[0] for i in range(3)]
            self.a_space = [g(f'app_plane_space.{i}')[0] for i in range(3)]
            self.a_time = [g(f'app_plane_time.{i}')[0] for i in range(3)]
            K, Fr = self.ds['num_keyframes'], self.ds['num_frames']
            self.tsf = (Fr - 1) / Fr                     # tensorf_dynamic.py:58-59
            self.tpo = 0.5 / K
        else:
            self.d_plane = [g(f'density_plane.{i}')[0] for i in range(3)]
            self.d_line = [g(f'density_line.{i}')[0] for i in range(3)]
            self.a_plane = [g(f'app_plane.{i}')[0] for i in range(3)]
            self.a_line = [g(f'app_line.{i}')[0] for i in range(3)]
        self.basis = g('basis_mat.weight')

    MAT = [[0, 1], [0, 2], [1, 2]]
    VEC = [2, 1, 0]
    MAT_T = [[2, 3], [1, 3], [0, 3]]

    def _feat_static(self, planes, lines, p):            # tensorf_no_sample.py:47-80 / 90-126
        out = []
        for i in range(3):
            pc = grid_sample_2d(planes[i], p[:, self.MAT[i][0]], p[:, self.MAT[i][1]])
            lc = grid_sample_2d(lines[i], np.zeros_like(p[:, 0]), p[:, self.VEC[i]])
            out.append(pc * lc)
        return np.concatenate(out, 0)                    # (sum C, N)

    def _feat_video(self, space, time, p):               # tensorf_dynamic.py:287-371
        out = []
        for i in range(3):
            if self.d_space[i].shape[0] == 0:            # skip test reads the *density* plane (:310, :355)
                continue
            sc = grid_sample_2d(space[i], p[:, self.MAT[i][0]], p[:, self.MAT[i][1]])
            tc = grid_sample_2d(time[i], p[:, self.MAT_T[i][0]], p[:, self.MAT_T[i][1]])
            out.append(sc * tc)
        return np.concatenate(out, 0)

    def _density(self, feat):
        if self.act == 'softplus':
            z = feat + self.density_shift
            return np.where(z > 20, z, np.log1p(np.exp(np.minimum(z, F32(20))))).astype(F32)
        if self.act == 'relu':
            return np.maximum(feat, F32(0))
        if self.act == 'relu_abs':
            return np.abs(feat)
        raise NotImplementedError(self.act)

    def color(self, x):                                   # tensorf_no_sample.py:128-280 / tensorf_dynamic.py:645-839
        pts = x['points']
        B, Z = pts.shape[:2]
        dist = x['distances'].reshape(B, Z)
        deltas = np.concatenate([dist[:, 1:] - dist[:, :-1], np.full((B, 1), 1e10, F32)], 1).astype(F32)
        viewdirs = x['viewdirs']
        valid = ~(((self.aabb[0] > pts) | (pts > self.aabb[1])).any(-1)) & (dist > 0)
        pn = ((pts - self.aabb[0]) * self.inv_size - F32(1)).astype(F32)
        if self.video:
            tn = ((x['base_times'] * F32(self.tsf) + F32(self.tpo)) * F32(2) - F32(1)).astype(F32)
            pn = np.concatenate([pn, tn], -1)
        sigma = np.zeros((B, Z), F32)
        if valid.any():
            pv = pn[valid]
            if self.video:
                f = self._feat_video(self.d_space, self.d_time, pv).sum(0, dtype=F32)
            else:
                f = self._feat_static(self.d_plane, self.d_line, pv).sum(0, dtype=F32)
            if not self.video:                           # video overwrites weights with ones (:702-704)
                f = f * x['weights'].reshape(B, Z)[valid]
            sigma[valid] = self._density(f)
        # raw2alpha (tensorf_utils.py:242-253)
        alpha = (F32(1.0) - np.exp(-sigma * (deltas * self.distance_scale))).astype(F32)
        T = np.cumprod(np.concatenate([np.ones((B, 1), F32), F32(1.0) - alpha + F32(1e-10)], -1), -1, dtype=F32)
        weight = (alpha * T[:, :-1]).astype(F32)
        app = weight > self.thr
        rgb = np.zeros((B, Z, 3), F32)
        if app.any():
            pa = pn[app]
            if self.video:
                feat = self._feat_video(self.a_space, self.a_time, pa)
            else:
                feat = self._feat_static(self.a_plane, self.a_line, pa)
            feat = (feat.T @ self.basis.T).astype(F32)    # basis_mat, Linear(bias=False)
            if self.shading == 'RGB':                    # tensorf_utils.py:341-343
                col = sigmoid(feat)
            else:                                        # tensorf_utils.py:334-338
                sh = eval_sh_bases_deg2(viewdirs[app])[:, None]
                col = np.maximum(np.sum(sh * feat.reshape(-1, 3, 9), -1, dtype=F32) + F32(0.5), F32(0))
            rgb[app] = col
        seen = x.get('_extracted')
        has = lambda k: k in x and (seen is None or k in seen)
        if has('color_scale'):                           # tensorf_utils.py:267-273
            rgb = rgb * (x['color_scale'] + F32(1.0)) + x['color_shift']
        elif has('color_transform'):
            raise NotImplementedError('per-sample color_transform')
        rgb_map = np.sum(weight[..., None] * rgb, -2, dtype=F32)
        if self.white_bg:
            rgb_map = rgb_map + (F32(1.0) - weight.sum(-1, dtype=F32)[:, None])
        if has('color_scale_global'):                    # scale_shift_color_one, tensorf_utils.py:275-281: sample 0's head
            rgb_map = rgb_map * (x['color_scale_global'][:, 0, :] + F32(1.0)) + x['color_shift_global'][:, 0, :]
        elif has('color_transform_global'):              # transform_color_one, tensorf_utils.py:308-320
            T = x['color_transform_global'].reshape(B, -1, 3, 3)[:, 0]
            sh = x['color_shift_global'].reshape(B, -1, 3)[:, 0]
            rgb_map = np.stack([rgb_map[:, c] + np.sum(rgb_map * T[:, c, :], -1, dtype=F32) for c in range(3)], -1) + sh
        rgb_map = np.clip(rgb_map, F32(0), F32(1)).astype(F32)
        return {'rgb': rgb_map, 'sigma': sigma, 'alpha': alpha, 'render_weights': weight,
                'valid': valid, 'rgb_samples': rgb.astype(F32)}

    # ---- public -----------------------------------------------------------------------
    def render(self, rays, chunk=16384, keep=('rgb',)):
        """render_chunked (rendering.py:100-150).  keep='all' returns every intermediate."""
        rays = _f(rays)
        outs = {}
        for i in range(0, rays.shape[0], chunk):
            x = self.embed(rays[i:i + chunk])
            c = self.color(x)
            x.update(c)
            for k, v in x.items():
                if not isinstance(v, np.ndarray):
                    continue
                if keep == 'all' or k in keep:
                    outs.setdefault(k, []).append(v)
        return {k: np.concatenate(v, 0) for k, v in outs.items()}
