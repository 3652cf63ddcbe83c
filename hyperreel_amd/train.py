"""Training step on the HIP device.

What INRSystem.training_step (nlf/__init__.py:634-709) gets from torch.autograd for the reference, split the way the
hardware wants it:
  * ray parameterisation + positional encoding: hr_train_features (no parameters, no gradient);
  * the sample-prediction MLP: `HipLinear`, a torch.autograd.Function over hr_linear_forward / hr_linear_backward -- every
    Linear (+ LeakyReLU) and its dgrad / wgrad / bias gradient as split-precision MFMA GEMMs (csrc/train_gemm_kernel.hip)
    on the reference-named nn.Linear parameters, exactly BaseMLP.forward (nlf/nets/mlp.py:159-172); torch only concatenates
    the skip layer's input;
  * everything after the MLP (head activations, intersection, sort, contraction, offsets / flow, VM gather, density,
    compositing, colour): one hand-written HIP forward and one backward kernel behind `SampleStage`, a
    torch.autograd.Function over the C ABI (hr_train_forward / hr_train_backward).
Gradients arrive on the reference's own parameters (planes, lines, basis_mat, MLP weights), so the reference's optimizers,
schedulers and regularizers apply unchanged.  There is no CPU path: the tensors must live on the HIP device.
"""
import ctypes as C

import torch

from . import lib as _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _tensors_struct(groups, basis, table=None):
    """groups: 4 lists of 3 tensors (density a, density b, app a, app b) -> hr_train_tensors."""
    t = _lib.hr_train_tensors()
    for name, grp in zip(('density_a', 'density_b', 'app_a', 'app_b'), groups):
        arr = getattr(t, name)
        for j in range(3):
            arr[j] = grp[j].data_ptr() if grp[j].numel() > 0 else None
    t.basis = basis.data_ptr() if basis.numel() > 0 else None
    t.color_table = table.data_ptr() if table is not None and table.numel() > 0 else None
    return t


class SampleStage(torch.autograd.Function):
    """rgb = f(head; planes, lines, basis_mat), not clamped (training mode, tensorf_no_sample.py:246).
    Inputs after `white_bg`: basis_mat.weight, then the 12 grid tensors in the order density a[0..2], density b[0..2],
    app a[0..2], app b[0..2] (a = plane | plane_space, b = line | plane_time), then -- only for models with a per-camera
    colour table (ColorTransformEmbedding, read with dataset.val_all) -- its `color_embedding`."""

    fields_out = None      # set by forward() when `want_fields` was given: {'distances', 'points', 'render_weights'} of that call, detached

    @staticmethod
    def forward(ctx, handle, rays, head, white_bg, basis, *tensors):
        grids, table = tensors[:12], (tensors[12] if len(tensors) > 12 else None)
        L = _lib.load()
        dev = rays.device
        if dev.type != 'cuda' or head.device != dev:
            raise RuntimeError('SampleStage runs on the HIP device; there is no CPU path')
        rays, head = rays.contiguous().float(), head.contiguous().float()
        vals = [g.detach().contiguous() for g in grids]
        groups = [vals[0:3], vals[3:6], vals[6:9], vals[9:12]]
        params = _tensors_struct(groups, basis.detach().contiguous(), None if table is None else table.detach().contiguous())
        rgb = torch.empty((rays.shape[0], 3), dtype=torch.float32, device=dev)
        want = SampleStage.want_fields
        SampleStage.want_fields = None
        with torch.cuda.device(dev):
            if want:
                # the forward's own per-sample values (hr_train_forward_fields): what a field-consuming regulariser reads, without a
                # second pass through the inference kernels
                from .plan import hr_fields
                B, Z = rays.shape[0], int(want)
                f = hr_fields()
                out = {'distances': torch.zeros((B, Z), dtype=torch.float32, device=dev), 'points': torch.zeros((B, Z, 3), dtype=torch.float32, device=dev),
                       'render_weights': torch.zeros((B, Z), dtype=torch.float32, device=dev)}
                f.distances_dev, f.points_dev, f.weights_dev = out['distances'].data_ptr(), out['points'].data_ptr(), out['render_weights'].data_ptr()
                _lib.check(L.hr_train_forward_fields(handle, C.byref(params), _ptr(rays), _ptr(head), B, int(bool(white_bg)), _ptr(rgb), C.byref(f),
                                                     _stream(dev)), 'hr_train_forward_fields')
                SampleStage.fields_out = out
            else:
                _lib.check(L.hr_train_forward(handle, C.byref(params), _ptr(rays), _ptr(head), rays.shape[0], int(bool(white_bg)), _ptr(rgb),
                                              _stream(dev)), 'hr_train_forward')
        ctx.handle, ctx.white_bg = handle, int(bool(white_bg))
        ctx.has_table = table is not None
        ctx.save_for_backward(rays, head, basis, *tensors)
        return rgb

    poison_outputs = False # tests: gradient buffers start as NaN instead of uninitialised memory
    want_fields = None     # set to z_channels right before apply() to make that call produce fields_out (single-threaded host code)

    @staticmethod
    def backward(ctx, d_rgb):
        L = _lib.load()
        rays, head, basis, *tensors = ctx.saved_tensors
        grids, table = tensors[:12], (tensors[12] if ctx.has_table else None)
        dev = rays.device
        d_rgb = d_rgb.contiguous().float()
        d_head = torch.empty_like(head)
        # hr_train_backward writes every element of every gradient tensor (the re-layout of the packed texel gradients covers the whole
        # (C, H, W) tensors; basis_mat's and the colour table's accumulators are cleared by the library on the stream): no zero fill here --
        # thirteen fills of 48 MB per step otherwise.  `poison_outputs` (tests) fills them with NaN instead, which must not survive.
        fresh = (lambda t: torch.full_like(t, float('nan'), memory_format=torch.contiguous_format)) if SampleStage.poison_outputs else \
            (lambda t: torch.empty_like(t, memory_format=torch.contiguous_format))
        g_grids = [fresh(g) for g in grids]
        g_basis = fresh(basis)
        g_table = None if table is None else fresh(table)
        gt = _tensors_struct([g_grids[0:3], g_grids[3:6], g_grids[6:9], g_grids[9:12]], g_basis, g_table)
        if g_basis.numel() == 0:
            raise RuntimeError('basis_mat has no columns')
        with torch.cuda.device(dev):
            _lib.check(L.hr_train_backward(ctx.handle, _ptr(rays), _ptr(head), _ptr(d_rgb), rays.shape[0], ctx.white_bg, _ptr(d_head),
                                           C.byref(gt), _stream(dev)), 'hr_train_backward')
        return (None, None, d_head, None, g_basis, *g_grids) + (() if table is None else (g_table,))


class CoarseRows(torch.autograd.Function):
    """Coarse level of a point_prediction cascade (nlf/embedding/point.py:137-203): raw head of the ray MLP (B, Zc * Pc) ->
    the point MLP's input rows (B * Zc, row_dim), and the gradient back (hr_train_rows_forward / _backward)."""

    @staticmethod
    def forward(ctx, handle, rays, head, n_rows_per_ray, row_dim):
        L = _lib.load()
        dev = rays.device
        if dev.type != 'cuda' or head.device != dev:
            raise RuntimeError('CoarseRows runs on the HIP device; there is no CPU path')
        rays, head = rays.contiguous().float(), head.contiguous().float()
        rows = torch.empty((rays.shape[0] * int(n_rows_per_ray), int(row_dim)), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.hr_train_rows_forward(handle, _ptr(rays), _ptr(head), rays.shape[0], _ptr(rows), _stream(dev)), 'hr_train_rows_forward')
        ctx.handle = handle
        ctx.save_for_backward(rays, head)
        ctx.shape = tuple(rows.shape)
        return rows

    @staticmethod
    def backward(ctx, d_rows):
        L = _lib.load()
        rays, head = ctx.saved_tensors
        dev = rays.device
        d_rows = d_rows.contiguous().float()
        scratch = torch.empty(ctx.shape, dtype=torch.float32, device=dev)
        d_head = torch.empty_like(head)
        with torch.cuda.device(dev):
            _lib.check(L.hr_train_rows_backward(ctx.handle, _ptr(rays), _ptr(head), _ptr(d_rows), rays.shape[0], _ptr(scratch), _ptr(d_head),
                                                _stream(dev)), 'hr_train_rows_backward')
        return None, None, d_head, None, None


def row_features(hc, rows):
    """Input of a cascade's point MLP: the identity parameterisation + positional encoding of its rows (nlf/pe.py:53-66,
    210-221), the arithmetic of hr_ray_features in torch ops because the rows carry a gradient (their point columns)."""
    cols = []
    for g in range(hc.n_groups):
        pg = hc.groups[g]
        if pg.fn != 0:
            raise NotImplementedError('point_prediction with a non-identity parameterisation of its inputs')
        x = rows[:, pg.start:pg.end]
        out = [x] if (pg.pe_type in (0, 2) or not pg.pe_exclude_identity) else []
        if pg.pe_type == 1:                                   # windowed: [sin(all), cos(all)] per frequency
            f = 1.0
            for j in range(pg.pe_n_freqs):
                f = f * pg.pe_freq_mult
                w = float(pg.pe_weight[j])
                out += [w * torch.sin(pg.pe_base_mult * f * x), w * torch.cos(pg.pe_base_mult * f * x)]
        elif pg.pe_type == 2:                                 # basic: [x, sin(f_j x_i) (i-major), cos(f_j x_i)]
            fr = torch.tensor([pg.pe_freq_mult ** (j + 1) for j in range(pg.pe_n_freqs)], dtype=x.dtype, device=x.device)
            cur = (fr[None, None] * x[..., None]).reshape(x.shape[0], -1)
            out += [torch.sin(cur), torch.cos(cur)]
        cols.append(torch.cat(out, -1))
    return torch.cat(cols, -1)


def ray_features(handle, rays, mlp_in):
    """rays (B, ray_dim) -> MLP input (B, mlp_in): RayParam + positional encoding on the device."""
    L = _lib.load()
    rays = rays.contiguous().float()
    out = torch.empty((rays.shape[0], int(mlp_in)), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        _lib.check(L.hr_train_features(handle, _ptr(rays), rays.shape[0], _ptr(out), _stream(rays.device)), 'hr_train_features')
    return out


class HipLinear(torch.autograd.Function):
    """y = LeakyReLU_slope(x W^T + b) (slope < 0: no activation) and its backward on the matrix cores
    (hr_linear_forward / hr_linear_backward: three bf16 MFMA products per fp32 GEMM, fp32 accumulation)."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope):
        L = _lib.load()
        dev = x.device
        if dev.type != 'cuda' or weight.device != dev:
            raise RuntimeError('HipLinear runs on the HIP device; there is no CPU path')
        x = x.float()
        if x.stride(-1) != 1:
            x = x.contiguous()
        w, b = weight.detach().contiguous().float(), bias.detach().contiguous().float()
        rows, fin, fout = x.shape[0], x.shape[1], w.shape[0]
        y = torch.empty((rows, fout), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.hr_linear_forward(_ptr(x), x.stride(0), rows, fin, _ptr(w), _ptr(b), fout, float(slope), _ptr(y), fout, _stream(dev)),
                       'hr_linear_forward')
        ctx.slope = float(slope)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.load()
        x, w, y = ctx.saved_tensors
        dev = x.device
        dy = dy.contiguous().float()
        rows, fin, fout = x.shape[0], x.shape[1], w.shape[0]
        dx = torch.empty((rows, fin), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty((fout,), dtype=torch.float32, device=dev)
        ws = torch.empty((max(int(L.hr_linear_workspace(rows, fin, fout)) // 4, 1),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.hr_linear_backward(_ptr(x), x.stride(0), _ptr(w), _ptr(y) if ctx.slope >= 0 else C.c_void_p(0), fout, _ptr(dy), fout,
                                            rows, fin, fout, ctx.slope, _ptr(dx) if dx is not None else C.c_void_p(0), fin, _ptr(dw), _ptr(db),
                                            _ptr(ws), _stream(dev)), 'hr_linear_backward')
        return dx, dw, db, None


class HipMLP(torch.autograd.Function):
    """BaseMLP.forward (nlf/nets/mlp.py:159-172) of the ray MLP as ONE launch (hr_mlp_train_forward: the render path's six-layer MFMA
    kernel on the current parameter values, which also leaves every hidden layer's output in HBM), and its backward layer by layer on
    the matrix cores (hr_linear_backward, as HipLinear).  Inputs: the model handle, rays, the MLP's input features (for layer 0's
    weight gradient), the skip mask, then weight_0, bias_0, weight_1, ... as the reference names them."""

    @staticmethod
    def forward(ctx, handle, rays, feats, skip_mask, n_out, *params):
        L = _lib.load()
        dev = rays.device
        nl = len(params) // 2
        ws = [p.detach().contiguous().float() for p in params[0::2]]
        bs = [p.detach().contiguous().float() for p in params[1::2]]
        n, fin = rays.shape[0], feats.shape[1]
        hidden = ws[0].shape[0]
        # x[l]: the input of layer l.  x[0] = feats; x[l] = the output of layer l - 1, behind a copy of feats where layer l is a skip layer
        xs = [feats]
        acts, lds, offs = [], [], []
        for l in range(1, nl):
            skip = (skip_mask >> l) & 1
            # a skip layer's input row is [feats | hidden]; two pad columns in FRONT of it keep the hidden part (what the kernels read
            # and write in 16-byte pieces) on a 16-byte boundary
            lead = (fin + ((-fin) % 4)) if skip else 0
            buf = torch.empty((n, lead + hidden), dtype=torch.float32, device=dev)
            x = buf[:, lead - fin:] if skip else buf
            if skip:
                x[:, :fin] = feats
            xs.append(x)
            acts.append(buf); lds.append(lead + hidden); offs.append(lead)
        head = torch.empty((n, int(n_out)), dtype=torch.float32, device=dev)
        PV = C.c_void_p * nl
        wv, bv = PV(*[w.data_ptr() for w in ws]), PV(*[b.data_ptr() for b in bs])
        av = PV(*([a.data_ptr() for a in acts] + [0]))
        ldv = (C.c_int64 * nl)(*(lds + [0]))
        offv = (C.c_int32 * nl)(*(offs + [0]))
        with torch.cuda.device(dev):
            _lib.check(L.hr_mlp_train_forward(handle, wv, bv, _ptr(rays), n, av, ldv, offv, _ptr(head), _stream(dev)), 'hr_mlp_train_forward')
        ctx.skip_mask, ctx.nl, ctx.fin, ctx.hidden = int(skip_mask), nl, fin, hidden
        ctx.save_for_backward(*xs, *ws)
        return head

    @staticmethod
    def backward(ctx, dhead):
        L = _lib.load()
        nl, fin, hidden = ctx.nl, ctx.fin, ctx.hidden
        xs, ws = ctx.saved_tensors[:nl], ctx.saved_tensors[nl:]
        dev = dhead.device
        dy = dhead.contiguous().float()
        dy_ptr, ld_dy = dy.data_ptr(), dy.shape[1]
        grads = [None] * (2 * nl)
        rows = xs[0].shape[0]
        keep = []
        with torch.cuda.device(dev):
            for l in range(nl - 1, -1, -1):
                x, w = xs[l], ws[l]
                fout, fin_l = w.shape[0], w.shape[1]
                last = (l == nl - 1)
                # y of layer l (the LeakyReLU mask) = the hidden part of x[l + 1]
                if last:
                    y_ptr, ldy = 0, fout
                else:
                    nxt = xs[l + 1]
                    y_ptr, ldy = nxt.data_ptr() + 4 * (nxt.shape[1] - hidden), nxt.stride(0)
                dx = None
                if l > 0:                               # same row shape as x[l]: the hidden columns of dx stay 16-byte aligned
                    lead = x.stride(0) - hidden
                    dxb = torch.empty((rows, lead + hidden), dtype=torch.float32, device=dev)
                    dx = dxb[:, lead + hidden - fin_l:]
                dw = torch.empty_like(w)
                db = torch.empty((fout,), dtype=torch.float32, device=dev)
                wsb = torch.empty((max(int(L.hr_linear_workspace(rows, fin_l, fout)) // 4, 1),), dtype=torch.float32, device=dev)
                _lib.check(L.hr_linear_backward(_ptr(x), x.stride(0), _ptr(w), C.c_void_p(y_ptr), ldy, C.c_void_p(dy_ptr), ld_dy, rows, fin_l, fout,
                                                -1.0 if last else 0.01, _ptr(dx) if dx is not None else C.c_void_p(0),
                                                dx.stride(0) if dx is not None else fin_l, _ptr(dw), _ptr(db), _ptr(wsb), _stream(dev)),
                           'hr_linear_backward')
                grads[2 * l], grads[2 * l + 1] = dw, db
                keep.append((dx, wsb))
                if dx is not None:                      # upstream of layer l - 1: the hidden columns of dx (a skip layer's first fin columns belong to the input)
                    dy_ptr, ld_dy = dx.data_ptr() + 4 * (fin_l - hidden), dx.stride(0)
        return (None, None, None, None, None, *grads)


def mlp_forward_fused(handle, rays, feats, net, skip_mask, n_out):
    """mlp_forward below through HipMLP: one forward launch.  The caller checks that the model qualifies (hidden width 256, no cascade)."""
    n = len(net.layers)
    params = []
    for i, layer in enumerate(net.layers):
        lin = layer[0] if i < n - 1 else layer
        params += [lin.weight, lin.bias]
    return HipMLP.apply(handle, rays, feats, int(skip_mask), int(n_out), *params)


def mlp_forward(net, x, skip_mask):
    """BaseMLP.forward (nlf/nets/mlp.py:159-172) on the reference-named parameters: Linear + LeakyReLU(0.01), the
    input concatenated in front of the activations at the skip layers, no activation after the last Linear."""
    inp = x
    n = len(net.layers)
    for i, layer in enumerate(net.layers):
        lin = layer[0] if i < n - 1 else layer
        if (skip_mask >> i) & 1:
            x = torch.cat([inp, x], -1)
        x = HipLinear.apply(x, lin.weight, lin.bias, 0.01 if i < n - 1 else -1.0)
    return x


def grid_parameters(net):
    """The 12 grid tensors of a HostTensorVM in SampleStage's order."""
    if net.video:
        groups = (net.density_plane_space, net.density_plane_time, net.app_plane_space, net.app_plane_time)
    else:
        groups = (net.density_plane, net.density_line, net.app_plane, net.app_line)
    return [p for grp in groups for p in grp]


class PlaneReg(torch.autograd.Function):
    """(1, C, H, W) plane -> [sum of squared vertical differences, sum of squared horizontal differences, sum |x|] in one
    pass (hr_plane_reg_forward); the backward is one elementwise kernel driven by the upstream gradient on the device."""

    @staticmethod
    def forward(ctx, plane):
        L = _lib.load()
        if plane.device.type != 'cuda':
            raise RuntimeError('PlaneReg runs on the HIP device; there is no CPU path')
        x = plane.detach().contiguous().float()
        _, c, h, w = x.shape
        sums = torch.zeros(3, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.hr_plane_reg_forward(_ptr(x), c, h, w, _ptr(sums), _stream(x.device)), 'hr_plane_reg_forward')
        ctx.save_for_backward(x)
        return sums

    @staticmethod
    def backward(ctx, d_sums):
        L = _lib.load()
        (x,) = ctx.saved_tensors
        _, c, h, w = x.shape
        coef = d_sums.contiguous().float()
        grad = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(L.hr_plane_reg_backward(_ptr(x), c, h, w, _ptr(coef), _ptr(grad), _stream(x.device)), 'hr_plane_reg_backward')
        return grad


def tv_loss(plane, weight=1.0):
    """TVLoss.forward (nlf/regularizers/tensorf.py:19-31) of one plane."""
    b, c, h, w = plane.shape
    s = PlaneReg.apply(plane)
    return weight * 2 * (s[0] / (c * (h - 1) * w) + s[1] / (c * h * (w - 1))) / b


def l1_mean(plane):
    """torch.mean(torch.abs(plane)) (density_L1, nlf/nets/tensorf_base.py:1024-1035)."""
    return PlaneReg.apply(plane)[2] / plane.numel()
