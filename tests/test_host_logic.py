"""CPU tests of the host-side mirror: config surface, YAML -> hr_config compiler, C-ABI library
loading (no compute calls), error behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest

from hyperreel_amd import config as C
from hyperreel_amd import plan, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.reference
@pytest.mark.parametrize('name', C.MODEL_NAMES)
def test_builtin_model_groups_equal_shipped_yaml(name):
    ref = C.to_plain(C.load_model_yaml(f'/root/reference/conf/experiment/model/{name}.yaml'))
    assert C.to_plain(C.model_config(name)) == ref


def test_final_grid_sizes():
    # utils/tensorf_utils.py:65-68 on N_voxel_final
    want = {'donerf_sphere': [600, 600, 600], 'technicolor_z_plane': [1007, 1007, 503],
            'neural_3d_z_plane': [823, 617, 514], 'immersive_sphere': [640, 640, 640]}
    for k, v in want.items():
        assert C.final_grid_size(C.model_config(k)) == v


def test_epoch_to_iter_rewrite():
    cfg = C.model_config('donerf_sphere')
    C.epoch_to_iter(cfg, 4000)
    sig = cfg.embedding.embeddings.ray_prediction_0.outputs.sigma.activation
    assert sig.window_iters == 12000 and sig.wait_iters == 0 and sig.window_epochs == 3
    pe = cfg.embedding.embeddings.ray_prediction_0.params.ray.pe
    assert pe.max_freq_iter == 0


def test_yaml_loader_reads_exponent_floats(tmp_path):
    p = tmp_path / 'm.yaml'
    p.write_text('a: 1e-3\nb: 2.5e2\nc: [1, 2]\nd: hello\n')
    cfg = C.load_model_yaml(str(p))
    assert cfg.a == 1e-3 and cfg.b == 250.0 and cfg.c == [1, 2] and cfg.d == 'hello'


@pytest.mark.parametrize('name', C.MODEL_NAMES)
def test_compile_config_matches_oracle_constants(name):
    """The flat hr_config and the oracle derive the same anchors / scales / contraction constants
    from the YAML by independent code."""
    from hyperreel_oracle import HyperReelOracle
    cfg, ds = C.model_config(name), C.dataset_scalars(name)
    grid = [20, 24, 28]
    sd = scenes.make_state_dict(cfg, ds, grid, seed=1)
    orc = HyperReelOracle(cfg, ds, sd)
    hc = plan.compile_config(cfg, ds, grid)
    Z = hc.z_channels
    assert Z == orc.Z
    assert np.array_equal(np.asarray(hc.samples[:Z], np.float32), orc.samples)
    assert np.float32(hc.z_scale) == np.float32(orc.z_scale)
    assert np.float32(hc.near) == np.float32(orc.near)
    assert hc.preds_per_z == sum(orc.out_shapes)
    assert hc.mlp_in == orc.layers[0][0].shape[1]
    assert hc.mlp_layers == len(orc.layers)
    assert [hc.grid[i] for i in range(3)] == grid
    if hc.contract_type == 1:
        assert np.float32(hc.c_r0) == np.float32(orc.contract.r0)
        assert np.float32(hc.c_d_scale) == np.float32(1.0 / (1.0 - orc.contract.d0 / orc.contract.d1))
    assert hc.mlp_precision == 4          # hidden 256 -> HR_MLP_AUTO: the library picks f16x3 / bf16x3 from its activation-range calibration
    assert plan.compile_config(cfg, ds, grid, mlp_precision='fp32').mlp_precision == 0


def test_out_of_scope_keys_raise():
    ds = C.dataset_scalars('donerf')
    cfg = C.model_config('donerf_sphere')
    cfg.embedding.embeddings.ray_intersect_0.intersect.type = 'deformable_voxel_grid'
    with pytest.raises(NotImplementedError, match='deformable_voxel_grid'):
        plan.compile_config(cfg, ds, [8, 8, 8])
    cfg.embedding.embeddings.ray_intersect_0.intersect.type = 'voxel_grid'     # needs a 1-channel z_vals head
    with pytest.raises(ValueError, match='z_vals channels'):
        plan.compile_config(cfg, ds, [8, 8, 8])
    cfg = C.model_config('donerf_sphere')
    cfg.color.net.shadingMode = 'MLP_Fea'
    with pytest.raises(NotImplementedError, match='MLP_Fea'):
        plan.compile_config(cfg, ds, [8, 8, 8])
    cfg = C.model_config('donerf_sphere')
    cfg.embedding.embeddings.ray_prediction_0.outputs.sigma.activation = C.to_cfg({'type': 'softplus'})
    with pytest.raises(NotImplementedError, match='softplus'):
        plan.compile_config(cfg, ds, [8, 8, 8])
    cfg = C.model_config('donerf_sphere')
    cfg.embedding.embeddings.ray_prediction_0.net.hidden_channels = 128
    assert plan.compile_config(cfg, ds, [8, 8, 8]).mlp_precision == 0      # falls back to exact fp32 MFMA
    with pytest.raises(NotImplementedError):
        plan.compile_config(cfg, ds, [8, 8, 8], mlp_precision='bf16x3')


def test_library_loads_and_exports_every_declared_symbol():
    """No GPU needed: dlopen the in-tree library and resolve each symbol of the header."""
    from hyperreel_amd import build, lib
    build.build()
    handle = lib.load()
    header = open(os.path.join(ROOT, 'include', 'hyperreel_hip.h')).read()
    declared = set(re.findall(r'\b(hr_[a-z_0-9]+)\s*\(', header))
    declared -= {'hr_model'}
    assert declared, 'no declarations found'
    bound = {name for name, _, _ in lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(handle, name) is not None
    assert handle.hr_abi_version() == lib.ABI_VERSION
    assert handle.hr_sizeof_config() == ctypes.sizeof(plan.hr_config)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'hyperreel_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'hyperreel_oracle' not in src and 'import oracle' not in src, f


def test_shard_bounds():
    from hyperreel_amd.parallel import shard_bounds, shard_range
    assert shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    assert shard_bounds(640000, 8)[-1] == 640000 and shard_range(640000, 7, 8) == (560000, 640000)
    assert shard_bounds(3, 8)[-1] == 3


def test_the_c_abi_splits_a_frame_like_the_python_side():
    """hr_shard_range (what a non-Python host calls per rank) against parallel.shard_bounds: the same contiguous split for every
    world size, ragged or not; the gather's slot is rank 0's share (the largest)."""
    import ctypes
    from hyperreel_amd import lib as hlib
    from hyperreel_amd.parallel import shard_bounds
    L = hlib.load()
    first, count = ctypes.c_int64(), ctypes.c_int64()
    for n in (0, 1, 7, 640000, 640003, 1 << 33):
        for world in (1, 2, 3, 4, 8):
            b = shard_bounds(n, world)
            counts = []
            for rank in range(world):
                hlib.check(L.hr_shard_range(n, rank, world, ctypes.byref(first), ctypes.byref(count)), 'hr_shard_range')
                assert (first.value, first.value + count.value) == (b[rank], b[rank + 1]), (n, world, rank)
                counts.append(count.value)
            assert counts[0] == max(counts) and sum(counts) == n
    assert L.hr_shard_range(10, 2, 2, ctypes.byref(first), ctypes.byref(count)) != 0      # rank out of range: an error, not a range
    assert L.hr_shard_range(10, 0, 0, ctypes.byref(first), ctypes.byref(count)) != 0


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    """The drop-in boundary is a C ABI: the header compiles as C99 and a C program can call the library (no GPU needed
    for the calls made here)."""
    import subprocess
    from hyperreel_amd import build
    lib = build.build()
    exe = tmp_path / 'abi_check'
    src = os.path.join(ROOT, 'tests', 'c_abi', 'abi_check.c')
    subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), src, '-o', str(exe),
                    '-L', os.path.dirname(lib), '-lhyperreel_hip', '-Wl,-rpath,' + os.path.dirname(lib),
                    '-Wl,-rpath,/opt/rocm/lib'], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert 'error text:' in r.stdout


def test_cascade_creation_contract():
    """hr_model_create_cascade validates the pair before touching the device: usable without a GPU."""
    import ctypes as CT
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import Golden
    from hyperreel_amd import lib
    L = lib.load()
    g = Golden('sweep/technicolor_cascaded')
    coarse, fine = plan.compile_cascade(g.cfg, g.dataset, g.grid)
    assert fine.casc_in_z == coarse.z_channels == 8 and fine.z_channels == 32 and fine.casc_row_dim == 7
    h = CT.c_void_p()
    # a cascade's fine config is refused by the single-level constructor
    assert L.hr_model_create(CT.byref(fine), CT.byref(h)) == -1 and b'hr_model_create_cascade' in L.hr_last_error()
    bad = plan.hr_config.from_buffer_copy(fine)
    bad.casc_in_z = 4
    assert L.hr_model_create_cascade(CT.byref(coarse), CT.byref(bad), CT.byref(h)) == -1
    assert b'casc_in_z' in L.hr_last_error()
    bad = plan.hr_config.from_buffer_copy(fine)
    bad.ray_dim = 6
    assert L.hr_model_create_cascade(CT.byref(coarse), CT.byref(bad), CT.byref(h)) == -1
    rc = L.hr_model_create_cascade(CT.byref(coarse), CT.byref(fine), CT.byref(h))
    assert rc in (0, -3)                                   # created (GPU present) or "no HIP device"
    if rc == 0:
        L.hr_model_destroy(h)


def test_ease_value_and_windowed_pe_weights():
    """nlf/activations.py:462-496 and nlf/pe.py:166-208 restated in plan.py: the weights the host folds into hr_config."""
    ev = {'type': 'ease_value', 'start_value': 1.0, 'window_iters': 12000, 'wait_iters': 4000, 'activation': 'sigmoid'}
    assert plan.ease_weight(ev, None) == 1.0
    assert plan.ease_weight(ev, 0) == 0.0 and plan.ease_weight(ev, 4000) == 0.0
    assert plan.ease_weight(ev, 10000) == 0.5 and plan.ease_weight(ev, 16000) == 1.0 and plan.ease_weight(ev, 10**7) == 1.0
    zero = dict(ev, window_iters=0)
    assert plan.ease_weight(zero, 3999) == 0.0 and plan.ease_weight(zero, 4000) == 1.0
    with plan.at_iteration(10000):
        a = plan._act(ev)
    assert a.type == 1 and a.outer == 0.5 and a.add == 0.5
    a = plan._act(ev)
    assert a.outer == 1.0 and a.add == 0.0
    pe = {'type': 'windowed', 'n_freqs': 4, 'max_freq_iter': 8000, 'wait_iters': 0}
    assert plan.windowed_pe_weights(pe, None) == [1.0] * 4
    w = plan.windowed_pe_weights(pe, 3000)
    assert w[0] == 1.0 and abs(w[1] - 0.5) < 1e-12 and w[2] == 0.0 and w[3] == 0.0
    assert plan.windowed_pe_weights(pe, 8001) == [1.0] * 4
    # the converged configuration is what a model that was never told an iteration compiles to
    # (INRSystem rewrites *_epochs into *_iters before it builds the model, nlf/__init__.py:305-315)
    cfg, ds = C.epoch_to_iter(C.model_config('donerf_sphere'), 4000), C.dataset_scalars('donerf_sphere')
    a = bytes(plan.compile_config(cfg, ds, [8, 8, 8]))
    assert a == bytes(plan.compile_config(cfg, ds, [8, 8, 8], iteration=10**7)) != bytes(plan.compile_config(cfg, ds, [8, 8, 8], iteration=2000))


def test_the_models_aabb_buffer_not_the_yaml_drives_the_compiled_box():
    """TensorBase keeps `aabb` as a buffer that `shrink` replaces during training (nlf/nets/tensorf_base.py:1191-1232);
    checkpoints carry it, and the reference renders with the loaded value.  So does the compiled configuration."""
    import torch
    from hyperreel_amd.models import HipLightfieldModel
    cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf_sphere')
    m = HipLightfieldModel(cfg, dataset=ds, grid_size=[8, 8, 8])
    yaml_box = [v for row in C.to_plain(cfg.color.net.aabb) for v in row]
    assert list(m._compile([8, 8, 8])[1].aabb) == pytest.approx(yaml_box)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd['color_model.net.aabb'] = torch.tensor([[-1.5, -1.0, -0.5], [1.0, 1.5, 2.0]])
    m.load_state_dict({'model.' + k: v for k, v in sd.items()})
    # ... together with the alpha mask of a trained checkpoint, which this path never reads (tensorf_no_sample.py:171)
    sd['color_model.net.alphaMask.alpha_aabb'] = sd['color_model.net.aabb'].clone()
    sd['color_model.net.alphaMask.alpha_volume'] = torch.ones(1, 1, 4, 4, 4)
    m.load_state_dict({'model.' + k: v for k, v in sd.items()}, strict=True)
    assert set(m.color_model.net.alpha_mask_state) == {'color_model.net.alphaMask.alpha_aabb', 'color_model.net.alphaMask.alpha_volume'}
    # ... but the opt-in occupancy early-reject does: the checkpoint's mask becomes the net's volume / box (D, H, W)
    assert tuple(m.color_model.net.alpha_volume.shape) == (4, 4, 4) and float(m.color_model.net.alpha_volume.min()) == 1.0
    assert m.color_model.net.alpha_aabb.reshape(-1).tolist() == [-1.5, -1.0, -0.5, 1.0, 1.5, 2.0]
    fresh = HipLightfieldModel(cfg, dataset=ds, grid_size=[8, 8, 8])
    with pytest.raises(RuntimeError, match='no alpha mask'):
        fresh.set_occupancy(True)
    hc = m._compile([8, 8, 8])[1]
    assert list(hc.aabb) == [-1.5, -1.0, -0.5, 1.0, 1.5, 2.0]
    assert list(hc.inv_size) == pytest.approx([2 / 2.5, 2 / 2.5, 2 / 2.5])
