"""What the optimizer costs in the training step: torch.optim.Adam in its three implementations on the model's own parameters with synthetic
gradients.  python tools/adam_probe.py [model].  Measurement aid (GPU box)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn
name = sys.argv[1] if len(sys.argv) > 1 else 'donerf_sphere'
cfg, ds = C.model_config(name), C.dataset_scalars(name)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
fn = build_render_fn(cfg, dataset=ds, grid_size=grid)
fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
fn.train()
params = [p for p in fn.model.parameters() if p.requires_grad]
n = sum(p.numel() for p in params)
res = {'model': name, 'tensors': len(params), 'parameters': n, 'floor_us_at_8TBs': round(n * 28 / 8e12 * 1e6, 1)}
for p in params:
    p.grad = torch.randn_like(p) * 1e-3
for label, kw in (('foreach', dict(foreach=True)), ('single', dict(foreach=False)), ('fused', dict(fused=True))):
    try:
        opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, **kw)
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            opt.step()
        torch.cuda.synchronize()
        res[label + '_ms'] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
    except Exception as e:          # noqa: BLE001
        res[label + '_error'] = repr(e)[:200]
print(json.dumps(res))
