"""CPU check of the occupancy ("alpha mask") arithmetic (hyperreel_amd/csrc/hr_mask.h, compiled for the host) and of the
torch-side grid management built on it (HostTensorVM.updateAlphaMask / shrink) against what the REFERENCE's own
getDenseAlpha / updateAlphaMask / shrink produced (tests/golden/mask/*.npz, oracle/refgen/make_alpha_mask.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from hyperreel_amd import config as cfgmod
from hyperreel_amd import plan, scenes
from test_train_host import GridPlane, ht  # noqa: F401  (the host library fixture and the plane descriptor)

HERE = os.path.dirname(os.path.abspath(__file__))
FP = C.POINTER(C.c_float)
IPt = C.POINTER(C.c_int)


def load(case):
    z = np.load(os.path.join(HERE, 'golden', 'mask', case + '.npz'))
    r = json.loads(bytes(z['recipe']).decode())
    cfg, ds = cfgmod.model_config(r['model']), r['dataset']
    if 'thre' in r:
        cfg['color']['net']['alpha_mask_thre'] = r['thre']
    sd = scenes.make_state_dict(cfg, ds, r['grid'], r['seed'], 'dense', 1.0)
    if r.get('carve', True):
        sd = scenes.carve_density(sd)
    assert abs(scenes.state_dict_checksum(sd) - r['checksum']) <= 1e-6 * max(1.0, abs(r['checksum']))
    return z, r, cfg, ds, sd


def pack(sd, video, names=None):
    """Packed density texels of the three plane pairs (appearance is not read by the mask)."""
    NET = 'model.color_model.net.'
    a_name, b_name = ('density_plane_space', 'density_plane_time') if video else ('density_plane', 'density_line')
    planes, keep = (GridPlane * 3)(), []
    for j in range(3):
        da, db = np.asarray(sd[f'{NET}{a_name}.{j}'], np.float32)[0], np.asarray(sd[f'{NET}{b_name}.{j}'], np.float32)[0]
        g = planes[j]
        nd = da.shape[0]
        g.cd4, g.ca4 = (nd + 3) // 4, 0
        g.ah, g.aw, g.bh, g.bw = da.shape[1], da.shape[2], db.shape[1], db.shape[2]
        g.tex = 4 * g.cd4
        pa = np.zeros((g.ah, g.aw, max(g.tex, 1)), np.float32)
        pb = np.zeros((g.bh, g.bw, max(g.tex, 1)), np.float32)
        pa[..., :nd], pb[..., :nd] = da.transpose(1, 2, 0), db.transpose(1, 2, 0)
        g.a, g.b = pa.ctypes.data, pb.ctypes.data
        keep += [pa, pb]
    return planes, keep


def dense_alpha(ht, hc, planes, n, num_frames, prev=None, prev_aabb=None):
    out = np.zeros(tuple(n), np.float32)
    nn = (C.c_int * 3)(*n)
    if prev is None:
        ht.ht_dense_alpha(C.byref(hc), planes, nn, C.c_float(0.01), num_frames, None, None, None, out.ctypes.data_as(FP))
    else:
        prev = np.ascontiguousarray(prev, np.float32)
        pn = (C.c_int * 3)(prev.shape[2], prev.shape[1], prev.shape[0])
        box = np.ascontiguousarray(prev_aabb, np.float32).reshape(-1)
        ht.ht_dense_alpha(C.byref(hc), planes, nn, C.c_float(0.01), num_frames, prev.ctypes.data_as(FP), pn, box.ctypes.data_as(FP),
                          out.ctypes.data_as(FP))
    return out


@pytest.mark.parametrize('case', ['alpha_mask_static', 'alpha_mask_video', 'alpha_mask_video_open'])
def test_dense_alpha_update_and_shrink_match_the_reference(ht, case):
    from hyperreel_amd.models import HipLightfieldModel
    z, r, cfg, ds, sd = load(case)
    video = cfg.color.net.type == 'tensor_vm_split_time'
    F = int(ds['num_frames'])
    hc = plan.compile_config(cfg, ds, r['grid'])
    planes, keep = pack(sd, video)
    # 1. getDenseAlpha without a mask
    a1 = dense_alpha(ht, hc, planes, r['n1'], F)
    assert np.abs(a1 - z['alpha1']).max() <= 2e-7 and z['alpha1'].max() > 1e-3

    # 2. updateAlphaMask + shrink on the host model, fed with that alpha
    m = HipLightfieldModel(cfg, dataset=ds, grid_size=r['grid'])
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=False)
    net = m.color_model.net
    net.getDenseAlpha = lambda n: torch.from_numpy(a1)          # the HIP launch, replaced by the host build's result
    new_aabb = net.updateAlphaMask(r['n1'])
    assert np.array_equal(net.alpha_volume.numpy(), z['mask_volume'])
    assert np.abs(new_aabb.numpy() - z['new_aabb']).max() <= 1e-6
    net.shrink(new_aabb)
    assert np.abs(net.aabb.numpy() - z['aabb_after']).max() <= 1e-6
    assert net.gridSize.tolist() == z['grid_after'].tolist() == m.grid_size
    own = dict(net.state_dict())
    for k in z.files:
        if k.startswith('after.'):
            assert np.array_equal(own[k[len('after.'):]].numpy(), z[k]), k

    # 3. getDenseAlpha again on the shrunk grid: the mask now rejects points (compute_alpha, tensorf_base.py:491-503)
    sd2 = {'model.color_model.net.' + k: v.numpy() for k, v in own.items()}
    hc2 = m._compile(m.grid_size)[1]
    planes2, keep2 = pack(sd2, video)
    a2 = dense_alpha(ht, hc2, planes2, r['n2'], F, prev=net.alpha_volume.numpy(), prev_aabb=net.alpha_aabb.numpy())
    assert np.abs(a2 - z['alpha2']).max() <= 2e-7
    assert ((a2 > 0) == (z['alpha2'] > 0)).all()
    if case == 'alpha_mask_video_open':
        # the fixture that tells the keyframe rule from the static one: the stored mask has holes inside its box (it keeps 58 %
        # of an uncarved scene), yet TensorVMKeyframeTime.compute_alpha never consults it -- alpha everywhere
        assert 0.3 < float(net.alpha_volume.numpy().mean()) < 0.8 and (z['alpha2'] > 0).all()
    else:
        assert 0.2 < (a2 > 0).mean() < 0.8


def test_set_iter_schedules_mask_shrink_and_growth_in_train_mode():
    """TensorBase.set_iter (tensorf_base.py:510-552): nothing in eval mode; in train mode the mask is rebuilt at the
    update_AlphaMask_list iterations (shrink at the first only) and the grids grow at the upsamp_list iterations to the
    log-linear N_voxel_list resolutions."""
    from hyperreel_amd.models import HipLightfieldModel
    cfg, ds = cfgmod.model_config('donerf_sphere'), cfgmod.dataset_scalars('donerf_sphere')
    m = HipLightfieldModel(cfg, dataset=ds)
    net = m.color_model.net
    assert net.update_alpha_mask_list == [4000, 8000] and net.upsamp_list == [4000, 6000, 8000, 10000, 12000]
    calls = []
    net.updateAlphaMask = lambda reso: calls.append(('mask', tuple(reso))) or torch.zeros(2, 3)
    net.shrink = lambda box: calls.append(('shrink',))
    net.upsample_volume_grid = lambda reso: calls.append(('grow', tuple(reso)))
    m.eval()
    m.set_iter(4000)
    assert calls == []
    m.train()
    for it in (3999, 4000, 4001, 6000, 8000):
        m.set_iter(it)
    g0 = tuple(m.grid_size)
    kinds = [c[0] for c in calls]
    assert kinds == ['mask', 'shrink', 'grow', 'grow', 'mask', 'grow']
    assert calls[0][1] == g0 and calls[4][1] == g0            # below 200 per axis: the mask lattice is the grid itself
    grown = [c[1] for c in calls if c[0] == 'grow']
    assert all(a[i] <= b[i] for a, b in zip(grown, grown[1:]) for i in range(3)) and grown[0][0] > g0[0]


@pytest.mark.reference
@pytest.mark.parametrize('name', ['donerf_sphere', 'technicolor_z_plane'])
def test_growth_schedule_equals_the_reference(name):
    import sys
    sys.path.insert(0, os.path.join(HERE, '..', 'oracle', 'refgen'))
    import ref_shim
    from hyperreel_amd.models import HipLightfieldModel
    ds = cfgmod.dataset_scalars(name)
    ref = ref_shim.build_reference(ref_shim.load_model_cfg(name), ds).model.color_model.net
    m = HipLightfieldModel(cfgmod.model_config(name), dataset=ds)
    net = m.color_model.net
    assert net.N_voxel_list == ref.N_voxel_list and net.upsamp_list == list(ref.upsamp_list)
    assert net.update_alpha_mask_list == list(ref.update_AlphaMask_list)
    assert m.grid_size == [int(v) for v in ref.gridSize.tolist()]
