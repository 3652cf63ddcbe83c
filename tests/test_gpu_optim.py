"""hyperreel_amd.optim.HipAdam (hr_adam_step) against torch.optim.Adam, the optimizer the reference builds (utils/__init__.py:49-76: eps 1e-8,
betas (0.9, 0.99), per-group lr / weight_decay)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed, sizes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.3).cuda()) for s in sizes]


SIZES = [(1,), (3,), (4097,), (256, 256), (7, 4096), (1, 8, 37, 41), (1000003,)]


@pytest.mark.parametrize('weight_decay', [0.0, 0.01])
def test_hip_adam_follows_torch_adam_step_for_step(weight_decay):
    from hyperreel_amd.optim import HipAdam
    a, b = _params(3, SIZES), _params(3, SIZES)
    groups = lambda ps: [{'params': ps[:3], 'lr': 1e-2}, {'params': ps[3:], 'lr': 2e-3, 'weight_decay': weight_decay}]
    ref = torch.optim.Adam(groups(a), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, foreach=False)
    hip = HipAdam(groups(b), lr=1e-3, betas=(0.9, 0.99), eps=1e-8)
    gen = torch.Generator().manual_seed(11)
    for step in range(6):
        for pa, pb in zip(a, b):
            g = (torch.randn(pa.shape, generator=gen) * (10.0 ** (-step))).cuda()          # gradients over six decades
            pa.grad, pb.grad = g.clone(), g.clone()
        ref.step()
        hip.step()
        for i, (pa, pb) in enumerate(zip(a, b)):
            d = (pa - pb).abs().max().item()
            assert d <= 2e-6 * max(1.0, pa.abs().max().item()), f'step {step} tensor {i}: {d:.3e}'
            sa, sb = ref.state[pa], hip.state[pb]
            assert float(sa['step']) == float(sb['step']) == step + 1
            for key in ('exp_avg', 'exp_avg_sq'):
                assert (sa[key] - sb[key]).abs().max().item() <= 2e-6 * sa[key].abs().max().item(), (step, i, key)


def test_hip_adam_state_dict_loads_into_torch_adam_and_back():
    from hyperreel_amd.optim import HipAdam
    a, b = _params(5, SIZES[:4]), _params(5, SIZES[:4])
    hip = HipAdam(a, lr=1e-3, betas=(0.9, 0.99))
    for p in a:
        p.grad = torch.ones_like(p) * 0.1
    hip.step()
    ref = torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.99), foreach=False)
    import copy
    ref.load_state_dict(copy.deepcopy(hip.state_dict()))          # (torch keeps the `step` tensors it is handed: without the copy both optimizers would count on one)
    for pa, pb in zip(a, b):
        pb.data.copy_(pa.data)
        pa.grad = torch.full_like(pa, -0.2)
        pb.grad = pa.grad.clone()
    hip.step()
    ref.step()
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-9)
    hip2 = HipAdam(_params(5, SIZES[:4]), lr=1e-3, betas=(0.9, 0.99))
    hip2.load_state_dict(copy.deepcopy(ref.state_dict()))
    assert float(next(iter(hip2.state.values()))['step']) == 2.0


def test_hip_adam_skips_parameters_without_gradient_and_refuses_the_cpu():
    from hyperreel_amd.optim import HipAdam
    a = _params(7, [(5,), (6,)])
    a[0].grad = torch.ones_like(a[0])
    before = a[1].detach().clone()
    opt = HipAdam(a, lr=1e-2)
    opt.step()
    assert torch.equal(a[1], before) and len(opt.state[a[1]]) == 0 and len(opt.state[a[0]]) == 3
    cpu = torch.nn.Parameter(torch.zeros(4))
    cpu.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match='HIP device'):
        HipAdam([cpu]).step()


def test_a_training_step_with_hip_adam_equals_one_with_torch_adam():
    """the whole step (forward_train, MSE, backward, optimizer) on a shipped family, deterministic gradient sums: after three steps the
    parameters agree with the torch.optim.Adam run to the optimizer's rounding"""
    from gpu_common import make_render_fn
    from helpers import Golden
    from hyperreel_amd.optim import HipAdam
    g = Golden('donerf_sphere_small')
    rays = torch.from_numpy(g.rays).cuda()
    target = torch.rand((rays.shape[0], 3), generator=torch.Generator().manual_seed(0)).cuda()
    finals = []
    for make in (lambda ps: torch.optim.Adam(ps, lr=1e-3, betas=(0.9, 0.99), eps=1e-8), lambda ps: HipAdam(ps, lr=1e-3, betas=(0.9, 0.99), eps=1e-8)):
        torch.manual_seed(0)                      # (parameters the fixture does not carry are drawn at construction)
        fn = make_render_fn(g.cfg, g.dataset, g.state_dict)
        fn.train()
        fn.model.set_train_deterministic(True)
        ps = [p for p in fn.model.parameters() if p.requires_grad]
        opt = make(ps)
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            ((fn.model.forward_train(rays, white_bg=False) - target) ** 2).mean().backward()
            opt.step()
        finals.append([p.detach().clone() for p in ps])
    for pa, pb in zip(*finals):
        assert (pa - pb).abs().max().item() <= 1e-5 * max(1.0, pa.abs().max().item())
