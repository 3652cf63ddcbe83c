"""CPU check of the training path's per-ray forward + backward (hyperreel_amd/csrc/hr_train.h, compiled for the host)
against torch.autograd on the CPU restatement of the reference (oracle/torch_port.py): gradients with respect to the
raw MLP output, every plane / line and basis_mat, on the five benchmark families.  The device build wraps the same
source in one thread per ray; the `-m gpu` test checks that build against this one's expectations."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from helpers import Golden, build_host_lib, trainable_sweep_cases
from hyperreel_amd import plan
from torch_port import TorchPort

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'host_math', 'hr_train_host.cpp')
OUT = os.path.join(HERE, 'host_math', '_build', 'libhr_train_host.so')
CSRC = os.path.join(HERE, '..', 'hyperreel_amd', 'csrc')

FP = C.POINTER(C.c_float)
MAT, VEC = [(0, 1), (0, 2), (1, 2)], [2, 1, 0]


class GridPlane(C.Structure):          # mirrors HrGridPlane (hyperreel_amd/csrc/hr_grid.h)
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p)] + [(k, C.c_int) for k in
                ('tex', 'aw', 'ah', 'bw', 'bh', 'cd4', 'ca4', 'ax', 'ay', 'bx', 'app_off', 'app_real', 'app_real_off')]


@pytest.fixture(scope='module')
def ht():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ('hr_train.h', 'hr_mask.h', 'hr_math.h', 'hr_grid.h')] + [os.path.join(HERE, '..', 'include', 'hyperreel_hip.h')]
    build_host_lib(OUT, SRC, deps)
    lib = C.CDLL(OUT)
    lib.ht_unsupported.restype = C.c_char_p
    assert lib.ht_sizeof_plane() == C.sizeof(GridPlane)
    return lib


def pack_grids(port, video):
    """Texel layout of hr_model_finalize (api.hip): [H][W][4*cd4 density | 4*ca4 appearance] per plane pair."""
    planes, packed, app_off, real_off = (GridPlane * 3)(), [], 0, 0
    for j in range(3):
        da, db, aa, ab = [t.detach().numpy()[0] for t in (port.d_a[j], port.d_b[j], port.a_a[j], port.a_b[j])]
        nd, na = da.shape[0], aa.shape[0]
        if video and nd == 0:
            na = 0
        g = planes[j]
        g.cd4, g.ca4 = (nd + 3) // 4, (na + 3) // 4
        g.ah, g.aw = da.shape[1], da.shape[2]
        g.bh, g.bw = db.shape[1], db.shape[2]
        g.app_off, g.app_real, g.app_real_off = app_off, na, real_off
        app_off += 4 * g.ca4
        real_off += na
        g.tex = 4 * (g.cd4 + g.ca4)
        pa = np.zeros((g.ah, g.aw, max(g.tex, 1)), np.float32)
        pb = np.zeros((g.bh, g.bw, max(g.tex, 1)), np.float32)
        pa[..., :nd] = da.transpose(1, 2, 0)
        pb[..., :nd] = db.transpose(1, 2, 0)
        pa[..., 4 * g.cd4:4 * g.cd4 + na] = aa[:na].transpose(1, 2, 0)
        pb[..., 4 * g.cd4:4 * g.cd4 + na] = ab[:na].transpose(1, 2, 0)
        g.a, g.b = pa.ctypes.data, pb.ctypes.data
        packed.append((pa, pb, nd, na, 4 * g.cd4))
    return planes, packed, app_off


# the five benchmark families, then every shipped model YAML (and variant) the training path accepts: z-plane / sphere /
# cylinder / voxel-grid / closest-point intersections, mip-NeRF-360 / bbox / z-depth / no contraction, static and keyframe
# grids, RGB and SH shading, per-sample and per-ray colour scale, ZeroMLP, unsorted / unmasked samples, and the
# activation / encoding / mask schedules inside their windows
CASES = ['donerf_sphere_small', 'donerf_cylinder_small', 'technicolor_z_plane_small', 'neural_3d_z_plane_small', 'immersive_sphere_small'] \
    + trainable_sweep_cases()


CASCADES = trainable_sweep_cases(cascades=True)


@pytest.mark.parametrize('white', [0, 1])
@pytest.mark.parametrize('case', CASES + CASCADES)
def test_backward_matches_autograd(ht, case, white):
    """Single-level models: the whole sample stage.  point_prediction cascades: their fine level, i.e. the same stage fed
    by the point MLP's head (one row of M samples per coarse point == (n, Z * P)), the coarse level held fixed."""
    g = Golden(case)
    cascade = plan.is_cascade(g.cfg)
    coarse_hc, hc = plan.compile_model(g.cfg, g.dataset, g.grid, iteration=g.iteration)
    assert ht.ht_unsupported(C.byref(hc)) is None
    port = TorchPort(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    n = min(192, g.rays.shape[0])
    rays = torch.from_numpy(np.ascontiguousarray(g.rays[:n], np.float32))
    grids = [t for grp in (port.d_a, port.d_b, port.a_a, port.a_b) for t in grp]
    table = None
    if hc.color_table_views > 0:                              # per-camera colour table (ColorTransformEmbedding, dataset.val_all)
        key = [k for k in g.state_dict if k.endswith('color_embedding')][0]
        table = port.color_table = torch.from_numpy(np.ascontiguousarray(g.state_dict[key], np.float32))
    leaves = grids + [port.basis] + ([table] if table is not None else [])
    for t in leaves:
        t.requires_grad_(True)

    def reference(rays):
        for t in leaves:
            t.grad = None
        with torch.no_grad():
            if cascade:                                       # the point MLP's own output, run once without a graph
                rec = {}
                orig = port._run_layers
                port._run_layers = lambda *a: rec.setdefault('h', orig(*a))
                port.embed(rays)
                port._run_layers = orig
                head0 = rec['h'].reshape(rays.shape[0], -1)
            elif port.o.zero_net:                             # ZeroMLP (nlf/nets/mlp.py:14-33): the head is all zeros
                head0 = torch.zeros(rays.shape[0], hc.z_channels * hc.preds_per_z)
            else:
                head0 = port._mlp(port._param_pe(rays))
        head = head0.clone().requires_grad_(True)
        x = port.embed(rays, point_head=head.view(rays.shape[0] * coarse_hc.z_channels, -1)) if cascade else port.embed(rays, head=head)
        rgb_ref = port.color(x, train=True, white_bg=bool(white))
        G = torch.randn(rgb_ref.shape, generator=torch.Generator().manual_seed(3))
        (rgb_ref * G).sum().backward()
        return head0, head, rgb_ref, G

    head0, head, rgb_ref, G = reference(rays)
    ok = torch.isfinite(head.grad).all(-1)
    if not bool(ok.all()):
        # torch.autograd returns NaN where a masked-out branch divides by zero (a ray along the cylinder axis: a == 0 in
        # intersect_utils.py:86-125; `torch.where` does not stop the NaN in backward).  The reference's own training step
        # would be poisoned by such a ray; the kernel's derivative of the taken branch is finite.  Compare on the others.
        assert int((~ok).sum()) <= 4
        rays = rays[ok]
        n = rays.shape[0]
        head0, head, rgb_ref, G = reference(rays)
        assert bool(torch.isfinite(head.grad).all())

    planes, packed, ca_total = pack_grids(port, port.o.video)
    gbuf = [(np.zeros_like(pa), np.zeros_like(pb)) for pa, pb, *_ in packed]
    g_a = (FP * 3)(*[b[0].ctypes.data_as(FP) for b in gbuf])
    g_b = (FP * 3)(*[b[1].ctypes.data_as(FP) for b in gbuf])
    basis = np.ascontiguousarray(port.basis.detach().numpy())
    d_basis = np.zeros_like(basis)
    rgb = np.zeros((n, 3), np.float32)
    d_head = np.zeros_like(head0.numpy())
    hnp, rnp, Gnp = np.ascontiguousarray(head0.numpy()), np.ascontiguousarray(rays.numpy()), np.ascontiguousarray(G.numpy())
    f = lambda a: a.ctypes.data_as(FP)
    tab = np.ascontiguousarray(table.detach().numpy()) if table is not None else None
    d_tab = np.zeros_like(tab) if tab is not None else None
    rc = ht.ht_train(C.byref(hc), f(rnp), f(hnp), C.c_longlong(n), f(Gnp), f(rgb), f(d_head), planes, g_a, g_b, f(basis), f(d_basis),
                     basis.shape[1], ca_total, white, f(tab) if tab is not None else None, f(d_tab) if tab is not None else None)
    assert rc == 0

    def close(got, ref, what):
        ref = np.asarray(ref, np.float64)
        scale = np.abs(ref).max()
        err = np.abs(np.asarray(got, np.float64) - ref).max()
        assert scale > 0, f'{what}: reference gradient is identically zero'
        assert err <= 2e-4 * scale + 1e-7, f'{what}: |err| {err:.3e} vs scale {scale:.3e}'

    ref_np = rgb_ref.detach().numpy()
    assert (np.abs(rgb - ref_np) <= 1e-5 * np.maximum(1.0, np.abs(ref_np))).all()      # unsorted / unmasked variants reach 1e5
    close(d_head, head.grad.numpy(), 'd head')
    P, ref_h = hc.preds_per_z, head.grad.numpy().reshape(n, hc.z_channels, -1)
    live = 0
    for col in range(P):                                   # every head column on its own scale (offsets, sigma, colour ...)
        if np.abs(ref_h[..., col]).max() > 0:
            close(d_head.reshape(ref_h.shape)[..., col], ref_h[..., col], f'd head column {col}')
            live += 1
        else:
            assert not d_head.reshape(ref_h.shape)[..., col].any()
    assert live >= 3
    close(d_basis, port.basis.grad.numpy(), 'd basis_mat')
    if table is not None:
        close(d_tab, table.grad.numpy(), 'd color_embedding')
    for j, (pa, pb, nd, na, aoff) in enumerate(packed):
        ga, gb = gbuf[j]
        for name, got, ref_t, cnt, off in (('density a', ga, port.d_a[j], nd, 0), ('density b', gb, port.d_b[j], nd, 0),
                                           ('app a', ga, port.a_a[j], na, aoff), ('app b', gb, port.a_b[j], na, aoff)):
            if cnt == 0:
                continue
            ref = ref_t.grad.numpy()[0][:cnt].transpose(1, 2, 0)
            close(got[..., off:off + cnt], ref, f'{name} {j}')
        pad = np.ones(ga.shape[-1], bool)
        pad[:nd] = False
        pad[aoff:aoff + na] = False
        assert not ga[..., pad].any() and not gb[..., pad].any()          # padding channels stay untouched


def test_unsupported_models_are_named(ht):
    g = Golden('donerf_sphere_small')
    hc = plan.compile_config(g.cfg, g.dataset, g.grid, grid_dtype='fp16')
    assert b'float16' in ht.ht_unsupported(C.byref(hc))
    assert ht.ht_unsupported(C.byref(plan.compile_config(g.cfg, g.dataset, g.grid))) is None


PIN = {'points': 0, 'viewdirs': 1, 'origins': 2, 'times': 3}      # HR_PIN_* (include/hyperreel_hip.h)


@pytest.mark.parametrize('case', CASCADES)
def test_cascade_rows_forward_and_backward_match_autograd(ht, case):
    """Coarse level of a point_prediction cascade (nlf/embedding/point.py:137-203): the rows handed to the point MLP and
    the gradient they send back to the ray MLP's raw head."""
    g = Golden(case)
    coarse, fine = plan.compile_model(g.cfg, g.dataset, g.grid, iteration=g.iteration)
    port = TorchPort(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    n = min(96, g.rays.shape[0])
    rays = torch.from_numpy(np.ascontiguousarray(g.rays[:n], np.float32))
    with torch.no_grad():
        head0 = torch.zeros(n, coarse.z_channels * coarse.preds_per_z) if port.o.zero_net else port._mlp(port._param_pe(rays))
    head = head0.clone().requires_grad_(True)
    rows_ref = port.embed(rays, head=head)['_rows']
    G = torch.randn(rows_ref.shape, generator=torch.Generator().manual_seed(5))
    (rows_ref * G).sum().backward()
    assert rows_ref.shape == (n * coarse.z_channels, fine.casc_row_dim)

    kinds = (C.c_int * 4)(*[fine.casc_input_kind[i] for i in range(4)])
    lens = (C.c_int * 4)(*[fine.casc_input_dim[i] for i in range(4)])
    rows = np.zeros(tuple(rows_ref.shape), np.float32)
    d_head = np.zeros_like(head0.numpy())
    f = lambda a: a.ctypes.data_as(FP)
    rnp, hnp, Gnp = np.ascontiguousarray(rays.numpy()), np.ascontiguousarray(head0.numpy()), np.ascontiguousarray(G.numpy())
    rc = ht.ht_rows(C.byref(coarse), f(rnp), f(hnp), C.c_longlong(n), f(Gnp), f(rows), f(d_head), fine.casc_row_dim, fine.casc_n_inputs, kinds, lens)
    assert rc == 0
    assert np.abs(rows - rows_ref.detach().numpy()).max() <= 2e-6
    ref = head.grad.numpy()
    if np.abs(ref).max() == 0:                   # a ZeroMLP ray level has a head nobody reads back
        assert not d_head.any()
        return
    P = coarse.preds_per_z
    ref_c, got_c = ref.reshape(n, coarse.z_channels, P), d_head.reshape(n, coarse.z_channels, P)
    live = 0
    for col in range(P):
        scale = np.abs(ref_c[..., col]).max()
        if scale > 0:
            assert np.abs(got_c[..., col] - ref_c[..., col]).max() <= 2e-4 * scale + 1e-7, col
            live += 1
        else:
            assert not got_c[..., col].any()
    assert live >= 1


@pytest.mark.parametrize('model,z', [('donerf_sphere', 96), ('technicolor_z_plane', 12), ('donerf_cylinder', 200)])
def test_backward_at_other_sample_counts(ht, model, z):
    """The per-ray arrays are sized by the next power of two of z_channels (8 ... 256): sample counts that are not powers
    of two, below and above one wavefront, against autograd (forward, head and plane gradients)."""
    from hyperreel_amd import config as cfgmod
    from hyperreel_amd import scenes
    cfg, ds = cfgmod.model_config(model, z_channels=z), cfgmod.dataset_scalars(model)
    grid = [24, 20, 16]
    sd = scenes.make_state_dict(cfg, ds, grid, seed=4, density='dense', app_scale=1.0)
    video = cfg.color.net.type == 'tensor_vm_split_time'
    if 'z_plane' in model:
        rays_np = scenes.random_rays(48, 2, video, pos_mean=(0, 0, 1.0), pos_std=0.15, dir_mean=(0, 0, -1.2), dir_std=0.5)
    else:
        rays_np = scenes.random_rays(48, 2, video)
    hc = plan.compile_config(cfg, ds, grid)
    assert hc.z_channels == z
    port = TorchPort(cfg, ds, sd)
    rays = torch.from_numpy(rays_np)
    grids = [t.requires_grad_(True) for grp in (port.d_a, port.d_b, port.a_a, port.a_b) for t in grp]
    port.basis.requires_grad_(True)
    with torch.no_grad():
        head0 = port._mlp(port._param_pe(rays))
    head = head0.clone().requires_grad_(True)
    rgb_ref = port.color(port.embed(rays, head=head), train=True, white_bg=True)
    G = torch.randn(rgb_ref.shape, generator=torch.Generator().manual_seed(1))
    (rgb_ref * G).sum().backward()
    planes, packed, ca_total = pack_grids(port, port.o.video)
    gbuf = [(np.zeros_like(pa), np.zeros_like(pb)) for pa, pb, *_ in packed]
    g_a = (FP * 3)(*[b[0].ctypes.data_as(FP) for b in gbuf])
    g_b = (FP * 3)(*[b[1].ctypes.data_as(FP) for b in gbuf])
    basis = np.ascontiguousarray(port.basis.detach().numpy())
    d_basis, rgb, d_head = np.zeros_like(basis), np.zeros((48, 3), np.float32), np.zeros_like(head0.numpy())
    f = lambda a: a.ctypes.data_as(FP)
    hnp, Gnp = np.ascontiguousarray(head0.numpy()), np.ascontiguousarray(G.numpy())
    assert ht.ht_train(C.byref(hc), f(rays_np), f(hnp), C.c_longlong(48), f(Gnp), f(rgb), f(d_head), planes, g_a, g_b, f(basis), f(d_basis),
                       basis.shape[1], ca_total, 1, None, None) == 0
    ref_np = rgb_ref.detach().numpy()
    assert (np.abs(rgb - ref_np) <= 1e-5 * np.maximum(1.0, np.abs(ref_np))).all()
    scale = np.abs(head.grad.numpy()).max()
    assert scale > 0 and np.abs(d_head - head.grad.numpy()).max() <= 2e-4 * scale
    ref_plane = port.d_a[0].grad.numpy()[0].transpose(1, 2, 0)
    got = gbuf[0][0][..., :ref_plane.shape[-1]]
    assert np.abs(got - ref_plane).max() <= 2e-4 * np.abs(ref_plane).max()
