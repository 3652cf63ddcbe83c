"""The operand roundings of the experimental f16f8 MLP arithmetic (csrc/mlp_f16f8_kernel.hip), emulated on the CPU
(tools/f16f8_emulation.py): what the choice of the two power-of-two operand scales buys.  Pins the design point the kernel and
hr_model_finalize implement (x_lo * 2^12, w_hi * 2^-12) -- measured on the device the same numbers come out
(profiles/r02_p_f16f8.txt: 1.4e-5 max, 2.5e-6 rms of max|head|)."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, ROOT)


def test_e4m3_rounding_table():
    import f16f8_emulation as E
    assert E.E4M3[-1] == 448.0 and E.E4M3[1] == 2.0 ** -9 and E.E4M3[8] == 2.0 ** -6
    x = np.array([0.0, 2.0 ** -10, 1.5 * 2.0 ** -9, 1.0, 1.0625, 1.1875, 447.0, 1e6, -3.3, 2.0 ** -6 * 1.0625], np.float64)
    want = np.array([0.0, 0.0, 2.0 ** -8, 1.0, 1.0, 1.25, 448.0, 448.0, -3.25, 2.0 ** -6], np.float64)   # ties go to the even mantissa
    assert np.array_equal(E.e4m3(x), want)


def test_scales_of_the_cross_terms():
    import f16f8_emulation as E
    from hyperreel_amd import config as C, scenes
    from hyperreel_oracle import HyperReelOracle
    name = 'donerf_sphere'
    cfg, ds = C.model_config(name), C.dataset_scalars(name)
    sd = scenes.make_state_dict(cfg, ds, [16, 16, 16], seed=7, density='dense', app_scale=1.0)
    rays = scenes.benchmark_rays(name, 800, 800, frame=7)
    r = np.ascontiguousarray(rays[np.random.default_rng(0).choice(rays.shape[0], 384, replace=False)])
    o = HyperReelOracle(cfg, ds, sd)
    x0 = o._param_pe(r)
    ref = E.run(o, x0, 'exact')
    scale = np.abs(ref).max()
    err = {m: np.abs(E.run(o, x0, m) - ref).max() / scale for m in ('f16x3', 'f16f8', 'f16x2', 'f16f8:0:0:6:-6')}
    assert err['f16x3'] < 2e-6
    assert err['f16f8'] < 3e-5 and err['f16f8'] < err['f16x2'] / 8          # the shipped scales: an order of magnitude inside f16x2
    assert err['f16f8:0:0:6:-6'] > 4 * err['f16f8']                          # x_lo in e4m3's subnormals: most of the gain is gone
