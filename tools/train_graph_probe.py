"""Is the training step host-bound?  The same optimizer step eagerly and as ONE replayed hipGraph (torch.cuda.graph: forward_train + loss + backward +
Adam(capturable)).  python tools/train_graph_probe.py [model] [batch].  Measurement aid (GPU box)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn

name = sys.argv[1] if len(sys.argv) > 1 else 'donerf_sphere'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
cfg, ds = C.model_config(name), C.dataset_scalars(name)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
fn = build_render_fn(cfg, dataset=ds, grid_size=grid, train_fused_mlp=bool(int(os.environ.get('HR_FUSED', '0'))))
fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
fn.train()
model = fn.model
rays_np = scenes.benchmark_rays(name, 800, 800, frame=7)
idx = np.random.default_rng(0).choice(rays_np.shape[0], batch, replace=False)
rays = torch.from_numpy(np.ascontiguousarray(rays_np[idx])).cuda()
target = torch.rand((batch, 3), device='cuda')
params = [p for p in model.parameters() if p.requires_grad]
if os.environ.get('HR_OPT', 'torch') == 'hip':                # the library's one-launch Adam (host-side step count: eager only, no graph figure)
    from hyperreel_amd.optim import HipAdam
    opt = HipAdam(params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8)
else:
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, capturable=True)
loss_out = torch.zeros((), device='cuda')


def step():
    opt.zero_grad(set_to_none=True)
    loss = ((model.forward_train(rays, white_bg=False) - target) ** 2).mean()
    loss.backward()
    opt.step()
    loss_out.copy_(loss.detach())


def timed(f, reps=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


def try_capture(label, f):
    import traceback
    try:
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(2):
                f()
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            f()
        torch.cuda.synchronize()
        print('capture', label, 'ok')
        return g_
    except Exception:          # noqa: BLE001
        tb = traceback.format_exc().strip().splitlines()
        print('capture', label, 'FAILED:', ' | '.join(l.strip() for l in tb)[-2500:])
        torch.cuda.synchronize()
        return None


if os.environ.get('HR_STAGES'):
    import ctypes
    from hyperreel_amd import lib as _hl
    hip = ctypes.CDLL('libamdhip64.so')
    _orig_check = _hl.check
    seen = {'bad': None}

    def cap_status():
        st = ctypes.c_int(0)
        hip.hipStreamIsCapturing(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(st))
        return st.value

    def check(rc, what=''):
        st = cap_status()
        if st == 2 and seen['bad'] is None:
            seen['bad'] = what
            print('capture invalidated at or before library call:', what)
        return _orig_check(rc, what)
    _hl.check = check
    import hyperreel_amd.train as _T
    import hyperreel_amd.models as _M
    _T._lib.check = check
    _M._lib.check = check

    from hyperreel_amd.train import ray_features
    keep = {}
    def f_feats():
        keep['f'] = ray_features(model._native, rays, model._hc.mlp_in)
    def f_fwd():
        with torch.no_grad():
            keep['o'] = model.forward_train(rays, white_bg=False)
    try_capture('forward_train (no grad)', f_fwd)
    def f_fb():
        for p_ in params:
            p_.grad = None
        ((model.forward_train(rays, white_bg=False) - target) ** 2).mean().backward()
    try_capture('forward + backward', f_fb)
    def f_opt():
        opt.step()
    f_fb()
    try_capture('adam step', f_opt)
    sys.exit(0)

res = {'model': name, 'batch': batch, 'eager_ms_per_step': round(timed(step), 4), 'loss_eager': float(loss_out)}
try:
    if os.environ.get('HR_OPT', 'torch') == 'hip':
        raise RuntimeError('HipAdam keeps its step count on the host: the eager figure is the one to read')
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    l0 = float(loss_out)
    res['graph_ms_per_step'] = round(timed(g.replay), 4)
    res['loss_first_replay'] = l0
    res['loss_after_replays'] = float(loss_out)
except Exception as e:          # noqa: BLE001
    res['graph_error'] = repr(e)[:600]
print(json.dumps(res))
