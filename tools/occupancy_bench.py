"""Speed of the opt-in occupancy early-reject on a sparse scene: the 600^3 DoNeRF benchmark scene with its density carved to
a sub-box (scenes.carve_density: what a trained scene looks like -- most of the volume empty), 800x800 frame, with and
without hr_model_set_occupancy.  python tools/occupancy_bench.py"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn

cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf_sphere')
sd = scenes.carve_density(scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0))
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
fn = build_render_fn(cfg, dataset=ds, grid_size=grid)
fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
rays = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 800, 800, frame=7)).cuda()
net = fn.model.color_model.net


def ms(n=30):
    for _ in range(5):
        fn.model.render(rays)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn.model.render(rays)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {'workload': 'donerf_sphere 600^3, density carved to a sub-box, 800x800 frame'}
plain = fn.model.render(rays)['rgb'].clone()
net.updateAlphaMask((200, 200, 200))
out['mask_kept_fraction'] = round(float(net.alpha_volume.mean()), 4)
for plan, fk in (('frame_kernel', True), ('two_kernels', False)):
    fn.model.set_execution(frame_kernel=fk)
    fn.model.set_occupancy(False)
    t_plain = ms()
    fn.model.set_occupancy(True)
    masked = fn.model.render(rays)['rgb'].clone()
    t_mask = ms()
    out[plan] = {'ms_per_frame_shipped': round(t_plain, 3), 'ms_per_frame_occupancy': round(t_mask, 3), 'speedup': round(t_plain / t_mask, 3),
                 'linf_masked_vs_shipped': float((masked - plain).abs().max())}
# the sample stage alone (what the compaction acts on): hr_stage_samples on the head of the frame's last chunk
import ctypes
from hyperreel_amd import lib as hlib
L = hlib.load(); h = fn.model.native()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rgb_tmp = torch.empty((rays.shape[0], 3), dtype=torch.float32, device='cuda')
chunk = 131072
def stage(f):
    for o in range(0, rays.shape[0], chunk):
        n = min(chunk, rays.shape[0] - o)
        hlib.check(f(o, n), 'stage')
def run_mlp(): stage(lambda o, n: L.hr_stage_mlp(h, ctypes.c_void_p(rays.data_ptr() + o * rays.shape[1] * 4), n, stream))
def run_smp(): stage(lambda o, n: L.hr_stage_samples(h, ctypes.c_void_p(rays.data_ptr() + o * rays.shape[1] * 4), n, ctypes.c_void_p(rgb_tmp.data_ptr() + o * 12), stream))
def ms_of(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
fn.model.set_execution(frame_kernel=False)
run_mlp()
fn.model.set_occupancy(False); fn.model.native(); t0_ = ms_of(run_smp)
fn.model.set_occupancy(True); fn.model.native(); t1_ = ms_of(run_smp)
out['sample_stage'] = {'ms_per_frame_shipped': round(t0_, 3), 'ms_per_frame_occupancy': round(t1_, 3), 'speedup': round(t0_ / t1_, 3)}
print(json.dumps(out))
