"""Pins the numpy oracle against outputs of the reference itself (tests/golden/*.npz,
made by oracle/refgen/make_golden.py).  CPU only."""
import numpy as np
import pytest

from helpers import Golden, trainable_sweep_cases, golden_cases, linf, sweep_cases, sweep_coverage
from hyperreel_oracle import HyperReelOracle


@pytest.mark.parametrize('case', golden_cases())
def test_oracle_matches_reference_golden(case):
    g = Golden(case)
    orc = HyperReelOracle(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    out = orc.render(g.rays, keep='all')
    # fp32 on both sides; only BLAS summation order / libm ulps differ.  On the 33 000-ray subsets of the full-size
    # white-noise grids (`*_full`: 640^3 / 823x617x514 texels of unit-variance noise) the last bits of a sample position
    # move the bilinear taps more than anywhere else: the tail of 33 000 rays reaches 3e-5 where 500 rays stay below 2e-5
    # `*_hostile`: MLP weights scaled x3 / x8 (scenes._hostile_layer) amplify the fp32 rounding of the matrix chain itself -- two correct
    # fp32 evaluations (numpy's BLAS order vs torch's) sit up to 4e-5 apart in RGB
    tol = 5e-5 if case in ('neural_3d_full', 'immersive_full') or case.endswith(('_hostile', '_stiff')) else 2e-5
    err = np.abs(out['rgb'] - g.rgb).max(-1)
    assert err.max() <= tol, f'{err.max():.3e}'
    assert (err > 2e-5).mean() <= (1e-2 if case.endswith(('_hostile', '_stiff')) else 1e-3)
    if 'distances' in g.arrays:
        n, Z = g.arrays['distances'].shape
        d_ref = g.arrays['distances']
        scale = 1.0 + np.abs(d_ref)
        assert float(np.max(np.abs(out['distances'].reshape(n, Z) - d_ref) / scale)) <= 2e-5
        p_ref = g.arrays['points']
        assert float(np.max(np.abs(out['points'] - p_ref) / (1.0 + np.abs(p_ref)))) <= 2e-5
        assert linf(out['render_weights'], g.arrays['render_weights']) <= 2e-5
        cs = g.arrays['color_scale']
        assert float(np.max(np.abs(out['color_scale'] - cs) / (1.0 + np.abs(cs)))) <= 2e-5
        if 'base_times' in g.arrays:
            assert linf(out['base_times'][:, 0, 0], g.arrays['base_times']) == 0.0


def test_golden_rgb_is_not_degenerate():
    for case in golden_cases():
        g = Golden.__new__(Golden)
        z = np.load(__import__('os').path.join(__import__('helpers').GOLDEN_DIR, case + '.npz'))
        assert z['rgb'].std() > 0.02 and np.isfinite(z['rgb']).all()


@pytest.mark.parametrize('case', golden_cases())
def test_torch_port_matches_reference_golden(case):
    """The multi-threaded torch-op port used as bench.py's CPU baseline computes the same image."""
    from torch_port import TorchPort
    g = Golden(case)
    out = TorchPort(g.cfg, g.dataset, g.state_dict, iteration=g.iteration).render(g.rays)
    assert linf(out['rgb'], g.rgb) <= (5e-5 if case.endswith(('_hostile', '_stiff')) else 2e-5)


@pytest.mark.parametrize('case', sweep_cases())
def test_oracle_matches_reference_on_every_accepted_shipped_yaml(case):
    g = Golden(case)
    out = HyperReelOracle(g.cfg, g.dataset, g.state_dict, iteration=g.iteration).render(g.rays)
    assert np.isfinite(g.rgb).all() and g.rgb.std() > 0.02
    assert linf(out['rgb'], g.rgb) <= 2e-5


def test_sweep_coverage_matches_the_plan_compiler():
    """coverage.json (what make_sweep.py saw) and the fixtures on disk agree; every rejected YAML
    carries a reason."""
    cov = sweep_coverage()
    have = {c.split('/', 1)[1] for c in sweep_cases()}
    assert have == {k for k, v in cov.items() if v['status'] == 'golden'}
    assert all(v.get('reason') for v in cov.values() if v['status'] != 'golden')
    assert len([k for k in have if not k.startswith('variant_')]) >= 45
    # nothing the reference can run is left out
    left = {k: v for k, v in cov.items() if v['status'] != 'golden' and v.get('reference_runs')}
    assert not left, left


@pytest.mark.parametrize('case', [c for c in sweep_cases() if 'iter' in c])
def test_schedule_fixtures_sit_inside_their_windows(case):
    """The *_iterN fixtures were rendered by the reference at training iteration N (model.set_iter(N)): the EaseValue /
    WindowedPE schedules must actually be active there, i.e. the converged model renders a different image."""
    g = Golden(case)
    assert g.iteration is not None
    converged = HyperReelOracle(g.cfg, g.dataset, g.state_dict).render(g.rays)['rgb']
    assert linf(converged, g.rgb) > 1.5e-4          # 10x what the oracle itself is held to


@pytest.mark.parametrize('case', trainable_sweep_cases() + trainable_sweep_cases(cascades=True))
def test_torch_port_matches_reference_on_the_trainable_shipped_yamls(case):
    """oracle/torch_port.py is the autograd reference of the training path's gradient checks: its forward is pinned here
    against what the reference itself rendered for every model family those checks use."""
    from torch_port import TorchPort
    g = Golden(case)
    out = TorchPort(g.cfg, g.dataset, g.state_dict, iteration=g.iteration).render(g.rays)
    assert linf(out['rgb'], g.rgb) <= 2e-5


GRAD_CASES = ['donerf_sphere_small', 'donerf_cylinder_small', 'technicolor_z_plane_small', 'neural_3d_z_plane_small', 'immersive_sphere_small']


@pytest.mark.parametrize('white', [0, 1])
@pytest.mark.parametrize('case', GRAD_CASES)
def test_torch_port_autograd_matches_the_reference_autograd(case, white):
    """The gradient comparator of the training path (torch.autograd on oracle/torch_port.py: tests/test_train_host.py,
    tests/test_gpu_train.py) is itself pinned to the REFERENCE'S OWN autograd: d sum(rgb * G) / d {planes, lines, basis_mat,
    every MLP weight and bias} from the reference modules in train mode (tests/golden/grad, oracle/refgen/make_grad_golden.py)."""
    import torch
    from helpers import GradGolden, port_leaves
    from torch_port import TorchPort
    g = Golden(case)
    gg = GradGolden(case, white)
    port = TorchPort(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    leaves = port_leaves(port)
    for t in leaves.values():
        t.requires_grad_(True)
    rays = torch.from_numpy(np.ascontiguousarray(g.rays[:gg.n_rays], np.float32))
    rgb = port.color(port.embed(rays), train=True, white_bg=bool(white))
    assert np.abs(rgb.detach().numpy() - gg.rgb).max() <= 2e-5
    (rgb * torch.from_numpy(gg.G)).sum().backward()
    checked = 0
    for name in gg.names():
        assert name in leaves, name
        t = leaves[name]
        if t.grad is None:
            continue
        gg.check(name, t.grad.numpy(), 2e-4)
        checked += 1
    assert checked >= 15
