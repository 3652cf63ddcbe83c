"""Shader clock and socket power while K1 runs back to back (rocm-smi polled from a thread), for the product and for measurement builds of it:
is K1's time an ENERGY figure?  (Every schedule change of round 6 -- prefetches, look-ahead 3 / 5 / 7, 4 or 8 wavefronts, staggered or
turn-taking workgroups -- ran level, while removing WORK (instructions, L2 traffic, MFMAs) always paid, additively.)
python tools/k1_power_probe.py [product k1_nomfma ...]"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib):
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    from hyperreel_amd import lib as hl
    if lib != 'product':
        hl.LIB_PATH = os.path.join(ROOT, 'tools', '_bin', f'libhr_{lib}.so')
    from hyperreel_amd import config as C, scenes
    from hyperreel_amd.render import build_render_fn
    cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf_sphere')
    stage = os.environ.get('HR_PROBE_STAGE', 'mlp')       # mlp (K1 back to back) | samples (K2) | frame (whole 800x800 frames through hr_render, default arithmetic)
    grid = None if stage in ('samples', 'frame') else [64, 64, 64]      # K2 wants the shipped 600^3 grid (its gather), K1 does not care
    sd = scenes.make_state_dict(cfg, ds, grid, seed=7, density='dense', app_scale=1.0)
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    f = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision='auto' if stage == 'frame' else 'f16f8')
    f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    h = f.model.native()
    rays_all = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 800, 800, frame=7)).cuda()
    rays = rays_all[:131072].contiguous()
    L = hl.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            try:
                d = json.loads(r.stdout)
                card = d[sorted(d)[0]]
                samples.append({k: v for k, v in card.items() if 'sclk' in k.lower() or 'power' in k.lower()})
            except Exception:
                samples.append({'raw': r.stdout[:200]})
            time.sleep(0.3)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    n = 0
    rgb = torch.empty((131072, 3), device='cuda')
    k2 = os.environ.get('HR_PROBE_STAGE') == 'samples'       # the sample stage instead (K2 reads the head K1 left in the workspace)
    hl.check(L.hr_stage_mlp(h, ctypes.c_void_p(rays.data_ptr()), 131072, st), 'hr_stage_mlp')
    rgb_all = torch.empty((rays_all.shape[0], 3), device='cuda')
    while time.perf_counter() - t0 < 5.0:
        for _ in range(200):
            if stage == 'frame':
                hl.check(L.hr_render(h, ctypes.c_void_p(rays_all.data_ptr()), rays_all.shape[0], ctypes.c_void_p(rgb_all.data_ptr()), st), 'hr_render')
            elif k2:
                hl.check(L.hr_stage_samples(h, ctypes.c_void_p(rays.data_ptr()), 131072, ctypes.c_void_p(rgb.data_ptr()), st), 'hr_stage_samples')
            else:
                hl.check(L.hr_stage_mlp(h, ctypes.c_void_p(rays.data_ptr()), 131072, st), 'hr_stage_mlp')
        torch.cuda.synchronize()
        n += 200
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    print(json.dumps({'variant': lib, 'stage': 'frame (hr_render, 640 000 rays)' if stage == 'frame' else ('K2' if k2 else 'K1'), 'ms_per_launch': round(dt / n * 1e3, 4), 'smi': samples[2:-1][:12]}))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(sys.argv[2])
    else:
        for v in (sys.argv[1:] or ['product']):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', v], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            print(next((ln for ln in r.stdout.splitlines() if ln.startswith('{')), f'{v}: failed rc {r.returncode}'), flush=True)
