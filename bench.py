#!/usr/bin/env python
"""Benchmark of the HyperReel forward-render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 without a launcher around it (no WORLD_SIZE in the environment): the script starts its N ranks itself -- torch.distributed.run,
one process per GPU, rendezvous on 127.0.0.1 -- and passes their one JSON line through.

Workload (BASELINE.json configs[1]): DoNeRF static scene, `donerf_sphere` model group, 800x800 pinhole frame = 640 000
rays, 32 samples/ray, shipped final grid 600^3, synthetic random-weight scene ('dense' density variant so alpha spans
(0,1)).  One step = one full forward render of a frame with rays and rgb resident in HBM.

Multi-GPU, both in every N > 1 line:
  `value` (scaling "weak", the driver's contract): every rank renders its own 800x800 frame (the same camera, panned by the rank) and the
      tiles are all-gathered over RCCL/xGMI inside the timed step; value = rays of all ranks per second.
  `strong` (BASELINE's "800x800 frame ms at 1/2/4/8", the north star's ">= 6x image-parallel scaling at 8 GPUs"): ONE 800x800 frame per
      step, split into N contiguous pixel ranges, the all-gather of frame i double-buffered under the render of frame i + 1
      (hyperreel_amd.parallel.ShardedFramePipeline).  frame_ms is the answer to "frame ms at N"; speedup_vs_n1_frame_ms prices it against
      a whole frame on one rank of the same job; tile_render_ms / gather_alone_ms / gather_overlap say where the rest went.
  `--scaling strong` makes the strong window `value` as well.

The JSON line also carries
  roofline        the dominant kernel of the step, its launches timed live with HIP events on the launch stream, against the
                  MI355X peak that bounds it (MI355X_MICROARCH.md: 2500 TFLOP/s dense bf16/fp16 MFMA);
  roofline_other  the other kernel: vector-ALU issue (wave-instructions per second against 1024 SIMDs x 2.4 GHz / 4 cycles);
                  the instruction count per launch comes from the committed PMC pass (the committed PMC pass);
  frame_kernel    the same frame through the single persistent frame kernel (head tile handed over in LDS);
  value_fp32_exact  the same frame with the exact fp32-MFMA MLP;
  value_f16x2     the same frame with the opt-in two-product MLP arithmetic, and its own parity against the oracle;
  value_f16f8     the same frame with the opt-in f16 + fp8 MLP arithmetic (one f16 product + one fp8 K=64 MFMA per 32 k), ditto;
  pytorch_gpu_baseline  the reference's algorithm as stock PyTorch-ROCm ops on this GPU (north star: ">= 10x");
  cpu_baseline    the CPU port of the reference's algorithm (oracle/torch_port.py) on a bounded sample of the same frame,
                  with its calibration against the reference itself (profiles/r02_cpu_calibration.json).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hyperreel_amd import config as C  # noqa: E402
from hyperreel_amd import scenes  # noqa: E402

# HR_BENCH_FAKE_RENDERER=1 (tests/test_bench_launch.py only): the launch / rank / window / JSON plumbing of a multi-rank run on CPU over gloo
# with a stand-in renderer -- no figure of such a run means anything, and its line says so in `data`
FAKE = os.environ.get('HR_BENCH_FAKE_RENDERER') == '1'
DEV = 'cpu' if FAKE else 'cuda'


def sync():
    if not FAKE:
        torch.cuda.synchronize()


MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 MFMA (= fp32 vector) peak
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (measured 2495)
VALU_PEAK_GINST = 1024 * 2.4 / 2.0   # wave-instructions per ns: 256 CUs x 4 SIMDs, a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md, wave
                                     # scheduling), 2.4 GHz.  tools/valu_ilp_ubench.hip on this part: 2.6 cycles per v_fma_f32 per SIMD with four wavefronts,
                                     # 4.3-4.9 per DPP instruction, 8.3 per transcendental (profiles/r03_valu_ilp_ubench.txt)
GATHER_UBENCH_L1_TBS, GATHER_UBENCH_L2_TBS = 30.2, 16.3      # profiles/r01_i_gather_ubench.txt (mode 1: quad-cooperative 64-byte texels)
HBM_PEAK_GBS = 8000.0


def algorithmic_bytes_per_ray(cfg, video, texel_bytes=4):
    """4*R_in + 12 + Z*(G_d + G_a), G = sum_i C_i * T * 4 bytes,
    T = 6 texels/channel (static VM: 4 plane + 2 line) or 8 (video: 4 space + 4 time)."""
    pred = cfg['embedding']['embeddings']['ray_prediction_0']
    Z = pred['z_channels']
    n = cfg['color']['net']
    T = 8 if video else 6
    nd, na = list(n['n_lamb_sigma']), list(n['n_lamb_sh'])
    if video:  # plane pairs without density components are skipped altogether
        na = [a if d > 0 else 0 for a, d in zip(na, nd)]
    G = (sum(nd) + sum(na)) * T * texel_bytes          # 'fp16 grids: halve the G terms' 
    return 4 * (8 if video else 6) + 12 + Z * G


def load_counters(suffix, matches):
    """The newest committed PMC counter file profiles/r0N_counters<suffix>.json whose workload matches -> (dict, 'fresh' | 'stale' |
    'none').  'fresh': collected from THIS tree's kernels (the file's csrc_hash equals hyperreel_amd.build.csrc_hash()); 'stale':
    another tree's -- the caller prints that and does not quote the numbers."""
    import glob
    from hyperreel_amd.build import csrc_hash
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r0?_counters{suffix}.json')), reverse=True)
    state = 'none'
    for f in files:
        try:
            tr = json.load(open(f))
            if not matches(tr['workload']):
                continue
        except (OSError, KeyError, ValueError):
            continue
        if tr.get('csrc_hash') == csrc_hash():
            return tr, 'fresh', os.path.relpath(f, ROOT)
        state = 'stale'
    return None, state, None


def mlp_flops_per_ray(cfg):
    return 2 * sum(o * i for o, i in scenes.mlp_layer_shapes(cfg))


def power_probe(fn, seconds=1.2, period=0.2):
    """Socket power and shader clock while `fn` (which enqueues GPU work) runs back to back: rocm-smi polled from a thread.  Returns
    {'socket_w', 'sclk_mhz', 'samples'} (medians) or None when rocm-smi is not usable.  Outside every timed region."""
    import json as _json, re, subprocess, threading
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5)
                card = _json.loads(r.stdout)
                card = card[sorted(card)[0]]
                w = next((float(v) for k, v in card.items() if 'power' in k.lower() and re.match(r'^[0-9.]+$', str(v))), None)
                c = next((float(re.sub(r'[^0-9.]', '', str(v))) for k, v in card.items() if k.lower().startswith('sclk clock speed')), None)
                if w is not None and c is not None:
                    samples.append((w, c))
            except Exception:
                return
            stop.wait(period)
    th = threading.Thread(target=poll, daemon=True)
    try:
        fn(); sync()
        th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                fn()
            sync()
    finally:
        stop.set()
        if th.is_alive():
            th.join(timeout=6)
    s_ = samples[1:] if len(samples) > 2 else samples          # (the first sample may predate the load)
    if not s_:
        return None
    med = lambda v: sorted(v)[len(v) // 2]
    return {'socket_w': med([a for a, _ in s_]), 'sclk_mhz': med([b for _, b in s_]), 'samples': len(s_)}


def time_stage(fn, reps):
    """Average ms of `fn()` (which enqueues on the current stream), HIP events around each call."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return float(np.mean(ts)), float(ts[0]), float(ts[len(ts) // 2])


def cpu_baseline(cfg, ds, sd, rays, n_sample):
    """The torch-op CPU port of the reference's algorithm (oracle/torch_port.py: the same ATen kernels the reference runs on
    a CPU -- grid_sample, cumprod, sort, addmm) on a bounded sample of the same frame.  Checker code, used here only as the
    reported CPU baseline; never on the measured path."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    from torch_port import TorchPort
    n_sample = min(n_sample, rays.shape[0])
    idx = np.arange(rays.shape[0]) if n_sample == rays.shape[0] else \
        np.sort(np.random.default_rng(0).choice(rays.shape[0], n_sample, replace=False))
    port = TorchPort(cfg, ds, sd)
    # torch's default of one thread per logical CPU is far from the optimum on many-core hosts (256-CPU GPU box: 16 threads
    # 0.17 Mrays/s, 128 threads 0.013): probe a few thread counts on a small slice and keep the fastest
    best = (0.0, torch.get_num_threads())
    for thr in sorted({8, 16, 32, min(64, os.cpu_count() or 8)}):
        if thr > (os.cpu_count() or 8):
            continue
        torch.set_num_threads(thr)
        port.render(rays[idx[:16384]])
        t0 = time.perf_counter()
        port.render(rays[idx[:32768]], chunk=16384)
        r = 32768 / (time.perf_counter() - t0)
        if r > best[0]:
            best = (r, thr)
    torch.set_num_threads(best[1])
    t0 = time.perf_counter()
    out = port.render(rays[idx], chunk=16384)
    dt = time.perf_counter() - t0
    return n_sample / dt, dt, idx, out['rgb']


def family_figures(names=('technicolor_z_plane', 'neural_3d_z_plane', 'immersive_sphere'), parity_rays=65536):
    """BASELINE configs[2..4]: the keyframe families' 800x800 frame at their shipped grids (synthetic seeded scenes), default
    arithmetic and execution plan, one hipGraph replay per frame; parity = rays over 1e-4 against the CPU port of the reference's
    algorithm on `parity_rays` rays of the same frame (checker code: timed nowhere)."""
    from hyperreel_amd.render import build_render_fn
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    from torch_port import TorchPort
    out = {}
    for name in names:
        cfg, ds = C.model_config(name), C.dataset_scalars(name)
        sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
        grid = [int(v) for v in sd['model.color_model.net.gridSize']]
        f = build_render_fn(cfg, dataset=ds, grid_size=grid)
        f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        rays_np = scenes.benchmark_rays(name, 800, 800, frame=7)
        rays = torch.from_numpy(rays_np).cuda()
        g0, _ = capture(f.model, rays)                                    # hr_render: rays at any times
        d0 = timed_frames(g0.replay, 20, 5, False, None)
        g, rgb = capture(f.model, rays, frame_time=float(rays_np[0, -1]))  # hr_render_frame: the frame's one time stated (what a viewer / video render knows)
        d = timed_frames(g.replay, 20, 5, False, None)
        g.replay()
        torch.cuda.synchronize()
        idx = np.sort(np.random.default_rng(0).choice(rays_np.shape[0], parity_rays, replace=False))
        ref = TorchPort(cfg, ds, sd).render(rays_np[idx], chunk=16384)['rgb']
        err = np.abs(rgb[torch.from_numpy(idx).cuda()].cpu().numpy() - ref).max(-1)
        Z = cfg['embedding']['embeddings']['ray_prediction_0']['z_channels']
        out[name] = {'value': round(rays_np.shape[0] / (d / 20) / 1e6, 3), 'unit': 'Mrays/s', 'ms_per_frame': round(d / 20 * 1e3, 4),
                     'entry': 'hr_render_frame (every ray of the frame at the frame\'s time: the keyframe row is read as a line)',
                     'ms_per_frame_hr_render': round(d0 / 20 * 1e3, 4),
                     'samples_per_ray': Z, 'grid': grid, 'mlp_gemm': f.model.mlp_precision_active(),
                     'execution': 'persistent frame kernel (head tile in LDS)' if f.model.frame_kernel_active() else
                                  'two kernels per workspace chunk (MLP -> HBM workspace -> sample stage)',
                     'parity_rays': int(parity_rays), 'parity_vs_oracle_linf': float(err.max()), 'parity_rays_over_1e-4': int((err > 1e-4).sum())}
        del f, g, g0, rgb, rays
        torch.cuda.empty_cache()
    return out


def timed_frames(step, steps, warmup, multi, dist):
    for _ in range(warmup):
        step()
    sync()
    if multi:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if multi:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def prewarm(step, seconds, multi, dist):
    """Replays `step` for about `seconds` of device time before anything is counted.  A fresh lease idles at 390 MHz and the power
    manager takes ~15 frames (30-40 ms) to reach the steady clock (profiles/r05_headline_diag_*.json: 2.35 -> 1.98 ms per frame over the
    first 15 replays of a process, again after 0.5 s of idling); 5 warm-up steps end inside that ramp.  Multi-rank: every rank runs the
    SAME number of steps (the step holds a collective)."""
    sync()
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    sync()
    per = max((time.perf_counter() - t0) / 8, 1e-5)
    n = int(min(max(seconds / per, 8), 4000))
    if multi:
        t = torch.tensor([n], dtype=torch.int64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = int(t.item())
    for i in range(n):
        step()
        if i % 32 == 31:
            sync()
    sync()
    return n + 8


def step_series(steps_by_name, n):
    """One HIP event pair around EVERY step, the plans interleaved (A, B, A, B, ...) in one window: {name: {min, p50, p90, max, mean}} in ms.
    Events are recorded on the stream the graphs replay on (torch's current stream)."""
    ev = {k: [] for k in steps_by_name}
    for _ in range(n):
        for k, st in steps_by_name.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); st(); b.record()
            ev[k].append((a, b))
    torch.cuda.synchronize()
    out = {}
    for k, pairs in ev.items():
        x = np.sort(np.asarray([a.elapsed_time(b) for a, b in pairs]))
        out[k] = {'min': round(float(x[0]), 4), 'p50': round(float(np.percentile(x, 50)), 4), 'p90': round(float(np.percentile(x, 90)), 4),
                  'max': round(float(x[-1]), 4), 'mean': round(float(x.mean()), 4), 'n': int(len(x))}
    return out


def capture(model, rays, frame_time=None):
    """One frame = hr_render's kernel launches, captured once into a hipGraph and replayed per step (the library neither
    allocates nor synchronises inside hr_render), so a slow host thread cannot starve the GPU between launches.
    frame_time: all rays carry this time (hr_render_frame)."""
    if frame_time is not None:
        render = lambda r: model.render(r, frame_time=frame_time)
    else:
        render = model.render
    render(rays)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        render(rays)                             # warm the side stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    # thread_local: only this thread's calls are policed during capture (the RCCL watchdog thread of a multi-rank run may
    # query its events meanwhile)
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        out = render(rays)['rgb']
    return graph, out


def viewer_figures(sizes=((512, 512), (800, 800)), model_name='immersive_sphere'):
    """BASELINE configs[4] (Google Immersive, viewer path): one displayed frame = camera -> rays on the device
    (hr_generate_rays; datasets/base.py:485-518) -> render -> the viewer's transpose / flip / 8-bit pack (hr_pack_display;
    utils/gui_utils.py:174-205), captured as ONE hipGraph and replayed; float16 texels, and next to the default f16x3 MLP the
    opt-in f16f8 and f16x2 arithmetics.  ms per displayed frame."""
    from hyperreel_amd.render import build_render_fn
    cfg, ds = C.model_config(model_name), C.dataset_scalars(model_name)
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    out = {'model': model_name, 'grid': grid, 'grid_dtype': 'fp16',
           'what': 'pose -> hr_generate_rays -> hr_render -> hr_pack_display (RGBA8, transposed + flipped like NeRFGUI.test_step), one hipGraph replay per frame'}
    pose = scenes.look_at_pose((0.3, 0.0, 0.0), (1.0, 0.1, 0.05))
    for prec in ('auto', 'f16x3', 'f16x2'):            # auto: the verified fast path (f16f8 + second pass)
        f = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=prec, grid_dtype='fp16')
        f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        m = f.model
        for (h, w) in sizes:
            foc = 0.5 * w / np.tan(0.5 * np.deg2rad(40.0))
            K = np.array([[foc, 0, w / 2], [0, foc, h / 2], [0, 0, 1]], np.float32)

            def frame():
                return m.pack_display(m.render_camera(pose, K, w, h, time=7.0 / 49.0), h, w, transpose=True, flip=True, rgba8=True)
            frame()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                frame()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                frame()
            d = timed_frames(g.replay, 30, 5, False, None)
            out[f'{"f16f8v" if prec == "auto" else prec}_{h}x{w}_ms'] = round(d / 30 * 1e3, 4)
        del f, m
        torch.cuda.empty_cache()
    return out


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU,
    `torch.distributed.run`, rendezvous on 127.0.0.1) with the same arguments, pass their output through, return their exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class _FakeModel:
    """HR_BENCH_FAKE_RENDERER: render(rays, out=None) -> {'rgb'} on the CPU, a pure function of the rays (so that gathered frames can be checked)."""

    def render(self, rays, out=None, **_):
        rgb = torch.sigmoid(rays[:, :3] * 3.0 + rays[:, 3:6])
        if out is not None:
            out.copy_(rgb)
            rgb = out
        return {'rgb': rgb}

    def mlp_overflowed(self): return False
    def mlp_precision_active(self): return 'fp32'
    def frame_kernel_active(self): return False
    def mlp_verified(self): return False


def local_frames(step, steps, warmup):
    """Seconds per step of `step` on THIS rank alone (no barrier, no reduction)."""
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return (time.perf_counter() - t0) / steps


def max_over_ranks(x, dist):
    t = torch.tensor([x], dtype=torch.float64, device=DEV)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def strong_window(model, args, dist, local_rank, headline, n1_frame_s):
    """ONE frame per step, split into contiguous pixel ranges (north star: "800x800 frame ms at 1/2/4/8", ">= 6x image-parallel scaling"):
    this rank renders its range into the pipeline's tile buffer, the all-gather of frame i runs on a side stream under the render of
    frame i + 1 (hyperreel_amd.parallel.ShardedFramePipeline).  Returns (dt, windows, n_pre, full frame, this rank's rays, info)."""
    from hyperreel_amd.parallel import ShardedFramePipeline
    rays_np = scenes.benchmark_rays(args.model, args.height, args.width, frame=7)
    n_pix = args.height * args.width
    pipe = ShardedFramePipeline(n_pix, torch.device(DEV, local_rank) if not FAKE else torch.device('cpu'), always_gather=True)
    rays = torch.from_numpy(np.ascontiguousarray(rays_np[pipe.lo:pipe.hi])).to(DEV)     # the rays generate_rays would produce
    model.render(rays)
    sync()
    if args.no_graph or FAKE:
        def step():
            tile = pipe.begin()
            model.render(rays, out=tile)
            return pipe.submit()
        tile_only = lambda: model.render(rays, out=pipe.tiles[0][:pipe.hi - pipe.lo])
    else:       # one hipGraph per tile buffer: a step is a graph launch + the all-gather enqueue
        step = pipe.capture(lambda tile: model.render(rays, out=tile))
        g_tile, _ = capture(model, rays)
        tile_only = g_tile.replay
    dt, windows, n_pre = headline(step)
    # host time per frame: how long the CPU needs to enqueue a step (no synchronisation inside the loop) -- the floor a rank's
    # frame time cannot go below however little of the frame it renders
    sync()
    t_h = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_us = (time.perf_counter() - t_h) / args.steps * 1e6
    sync()
    rgb_full = pipe.flush()
    sync()
    # what the pieces cost alone: this rank's tile without any collective, and the all-gather without any render
    tile_s = max_over_ranks(local_frames(tile_only, args.steps, args.warmup), dist)

    def gather_only():
        dist.all_gather_into_tensor(pipe.full[0], pipe.tiles[0])
    gather_s = max_over_ranks(local_frames(gather_only, args.steps, args.warmup), dist)
    frame_s = dt / args.steps
    exposed = max(0.0, frame_s - tile_s)
    info = {'what': f'ONE {args.height}x{args.width} frame per step split into {pipe.world} contiguous pixel ranges, all-gather of frame i under the render of frame i + 1',
            'frame_ms': round(frame_s * 1e3, 4), 'mrays_s': round(n_pix / frame_s / 1e6, 3),
            'n1_frame_ms': round(n1_frame_s * 1e3, 4),
            'speedup_vs_n1_frame_ms': round(n1_frame_s / frame_s, 3),
            'tile_render_ms': round(tile_s * 1e3, 4), 'gather_alone_ms': round(gather_s * 1e3, 4),
            'gather_overlap': round(min(1.0, max(0.0, 1.0 - exposed / gather_s)), 3) if gather_s > 0 else None,
            'host_us_per_frame': round(host_us, 1), 'ranks': pipe.world, 'rays_per_rank': int(pipe.hi - pipe.lo),
            'windows_ms_per_frame': [round(w / args.steps * 1e3, 4) for w in windows],
            'answers': 'BASELINE metric "800x800 frame ms at N GPUs" = frame_ms; north star ">= 6x image-parallel scaling at 8 GPUs" = speedup_vs_n1_frame_ms'}
    return dt, windows, n_pre, rgb_full, rays, pipe, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--model', default='donerf_sphere')
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=800)
    ap.add_argument('--chunk', type=int, default=0, help='rays per internal workspace chunk (0 = library default)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--cpu-sample', type=int, default=640000, help='rays of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--prewarm', type=float, default=0.5, help='seconds of uncounted replays before the first warm-up step (clock ramp of an idle GPU; 0 = none)')
    ap.add_argument('--windows', type=int, default=3, help='consecutive timed windows of exactly --steps steps each; value = their median')
    ap.add_argument('--no-stage-timing', action='store_true')
    ap.add_argument('--no-power', action='store_true', help='skip the rocm-smi power / clock probe of the two kernels (about 4 s)')
    ap.add_argument('--no-extras', action='store_true', help='skip frame_kernel / value_fp32_exact / pytorch_gpu_baseline')
    ap.add_argument('--mlp-precision', default='auto', choices=['auto', 'bf16x3', 'f16x3', 'f16f8', 'f16x2', 'fp32'],
                    help="arithmetic of the MLP GEMMs: auto = the library's choice (f16 + fp8 first pass with its discrete decisions verified by an f16x3 "
                         "second pass where that applies, DESIGN 3c; f16x3 / bf16x3 otherwise), or one arithmetic by name (fp32 = the exact fp32 MFMA)")
    ap.add_argument('--no-graph', action='store_true', help='enqueue every frame eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--grid-dtype', default='fp32', choices=['fp32', 'fp16'],
                    help='texel storage: float32 (the reference; headline) or float16 (viewer path, BASELINE config 5)')
    ap.add_argument('--lib', default='', help='measurement builds only (tools/build_variant.py): load this library instead of the in-tree one')
    ap.add_argument('--frame-kernel', action='store_true', help='opt-in plan: the persistent frame kernel, head tile in LDS')
    ap.add_argument('--no-frame-kernel', action='store_true', help='(default) two-kernel path through the HBM workspace')
    ap.add_argument('--frame-mode', type=int, default=1, choices=[1, 2], help='HR_OPT_FRAME_KERNEL: 1 = the frame kernel where it is the faster plan (static nets), 2 = wherever the model fits it (keyframe families on 32-ray tiles)')
    ap.add_argument('--sample-waves', type=int, default=0, choices=[0, 4, 8], help='sample wavefronts per workgroup of the frame kernel (0 = library default)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: be the launcher (the driver's own multi-GPU command comes in through torch.distributed.run with WORLD_SIZE set)
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: one process per GPU, launched as torch.distributed.run --nproc-per-node {args.gpus}')
    if not FAKE:
        torch.cuda.set_device(local_rank)
    dist = None
    # HR_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, all-gather, barrier, max-reduce, the strong window) with a single
    # rank -- the only way to exercise it on a one-GPU box
    multi = world > 1 or os.environ.get('HR_BENCH_FORCE_DIST') == '1'
    if multi:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if 'RANK' not in os.environ:       # HR_BENCH_FORCE_DIST=1 without a launcher: a one-rank job on the loopback
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
            os.environ.update({'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1'})
            os.environ.setdefault('MASTER_PORT', str(port))
        if FAKE:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    if args.lib:
        from hyperreel_amd import lib as _hl
        _hl.LIB_PATH = os.path.abspath(args.lib)
    from hyperreel_amd.render import build_render_fn
    cfg = C.model_config(args.model)
    ds = C.dataset_scalars(args.model)
    video = cfg['color']['net']['type'] == 'tensor_vm_split_time'
    sd = scenes.make_state_dict(cfg, ds, [16, 16, 16] if FAKE else None, seed=7, density='dense', app_scale=1.0)
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    texel_bytes = 2 if args.grid_dtype == 'fp16' else 4
    # checker-side weights: with float16 texels the reference algorithm is run on the same rounded grids
    sd_ref = sd if texel_bytes == 4 else {k: (v.astype(np.float16).astype(np.float32) if ('_plane' in k or '_line' in k) else v)
                                          for k, v in sd.items()}

    def make(precision, frame_kernel, grid_dtype=None):
        f = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=precision, grid_dtype=grid_dtype or args.grid_dtype,
                            frame_kernel=frame_kernel, sample_waves=args.sample_waves or None)
        f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        if args.chunk:
            f.model.reserve(args.chunk)
        f.model.native()
        return f

    def headline(step):
        """value's window: uncounted pre-warm by time, then --windows consecutive windows of W warm-up + EXACTLY K timed steps (barrier +
        synchronize on both sides, max over ranks); the median window is the figure, all of them are printed."""
        n_pre = prewarm(step, args.prewarm, multi, dist) if args.prewarm > 0 else 0
        wins = [timed_frames(step, args.steps, args.warmup, multi, dist) for _ in range(max(1, args.windows))]
        return sorted(wins)[(len(wins) - 1) // 2], wins, n_pre

    use_frame = (2 if args.frame_mode == 2 else True) if (args.frame_kernel and not args.no_frame_kernel) else False
    model = _FakeModel() if FAKE else make(args.mlp_precision, use_frame).model
    want_strong_value = args.scaling == 'strong' and multi
    Z = cfg['embedding']['embeddings']['ray_prediction_0']['z_channels']

    # ---- the weak window: every rank renders its own frame (the same camera, panned by the rank), the tiles all-gathered inside the step
    graph = None
    rays_np = scenes.benchmark_rays(args.model, args.height, args.width, frame=7 + rank)
    if world > 1:
        pose_shift = np.zeros_like(rays_np)
        pose_shift[:, 1] = 0.01 * rank
        rays_np = rays_np + pose_shift
    rays = torch.from_numpy(rays_np).to(DEV)
    B = rays.shape[0]
    gathered = torch.empty((world, B, 3), dtype=torch.float32, device=DEV) if multi else None
    if not args.no_graph and not FAKE:
        graph, rgb_static = capture(model, rays)

    def render_frame():
        if graph is not None:
            graph.replay()
            return rgb_static
        return model.render(rays)['rgb']

    def step():
        out = render_frame()
        if multi:
            dist.all_gather_into_tensor(gathered.view(-1), out.view(-1))
        return out

    strong_info = None
    if not want_strong_value:
        dt, windows, n_pre = headline(step)
        rgb = step()
        sync()
        total_rays = B * world
        rays_per_gpu = B
        parallelism = f'image tiles x{world}, RCCL all_gather of rgb' if world > 1 else 'single GPU'
    if multi:
        # ---- the strong window (north star's frame-ms metric): one frame split N ways.  Its yardstick is THIS job's single-GPU frame: a whole
        # frame on one rank without any collective (every rank times its own; the slowest counts)
        n1 = max_over_ranks(local_frames(render_frame, args.steps, args.warmup), dist)
        s_dt, s_windows, s_pre, rgb_full, s_rays, pipe, strong_info = strong_window(model, args, dist, local_rank, headline, n1)
        if want_strong_value:
            dt, windows, n_pre = s_dt, s_windows, s_pre
            rays, B = s_rays, s_rays.shape[0]
            rgb = rgb_full[pipe.lo:pipe.hi]
            total_rays = args.height * args.width
            rays_per_gpu = B
            parallelism = f'ONE {args.height}x{args.width} frame split into {world} pixel ranges, double-buffered RCCL all_gather'
            host_us = strong_info['host_us_per_frame']
            graph = None
    strong = want_strong_value

    ms_per_step = dt / args.steps * 1e3
    value = total_rays / (dt / args.steps) / 1e6
    # every step of one more window with its own event pair: a stall (clock dip, a neighbour's SMI poll) shows as max >> p50 in the line itself
    series = step_series({'value_path': step}, max(args.steps, 20)) if not FAKE else None
    # the fp16 split arithmetic was chosen on calibration rays: the kernels' sticky overflow bit must be clear on the rays that were timed
    # (render() checks it on the first batches and falls back to bf16x3, models.py _overflow_guard; a replayed graph cannot)
    if model.mlp_overflowed():
        raise SystemExit('bench: an MLP activation left the IEEE-half range on the benchmark rays (HR_OPT_MLP_OVERFLOW): the figure would be invalid')
    prec_name = model.mlp_precision_active()      # 'auto' resolved by the library's activation-range calibration (hr_model_finalize)
    result = {
        'metric': 'Mrays/s (32 samples/ray), forward render of 800x800 frames',
        'value': round(value, 3), 'unit': 'Mrays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
        'dtype': 'f32 storage / accumulate; MLP GEMM operands: see mlp_gemm', 'mlp_gemm': None,
        'data': 'synthetic (seeded random-weight scene, dense density variant; pinhole rays)' if not FAKE else 'NONE: HR_BENCH_FAKE_RENDERER (CPU stand-in renderer over gloo; launch plumbing test, no figure means anything)',
        'windows_ms_per_step': [round(w / args.steps * 1e3, 4) for w in windows],
        'value_rule': f'median of {len(windows)} consecutive windows of exactly {args.steps} steps (each after {args.warmup} warm-up steps), wall clock, max over ranks',
        'step_ms': series,
        'config': {'workload': f'BASELINE configs[1]: DoNeRF static ({args.model}), {args.height}x{args.width} frame '
                               f'= {args.height * args.width} rays, {Z} samples/ray, grid {grid[0]}x{grid[1]}x{grid[2]}, single forward render',
                   'rays_per_gpu': rays_per_gpu, 'parallelism': parallelism, 'frame_ms': round(ms_per_step, 4),
                   'execution': 'persistent frame kernel (head tile in LDS)' if model.frame_kernel_active() else
                                'two kernels per workspace chunk (MLP -> HBM workspace -> sample stage)'},
    }

    if strong_info is not None:
        strong_info['rccl_ranks'] = int(dist.get_world_size())
        strong_info['backend'] = str(dist.get_backend())
        result['strong'] = strong_info
    # ---- per-kernel timing + roofline (rank 0): the two kernels of the default path through hr_stage_*
    if rank == 0 and not args.no_stage_timing and not FAKE:
        import ctypes
        from hyperreel_amd import lib as hlib
        L = hlib.load()
        h = model.native()
        # rays per launch, as hr_render splits a call: as many launches as the workspace (163 840 rays by default) demands, of equal size
        try:
            cap = model.chunk_rays()
        except Exception:               # (a measurement library older than HR_OPT_CHUNK_RAYS, --lib)
            cap = args.chunk or 163840
        n_launch = -(-B // cap)
        chunk = min(cap, (-(-B // n_launch) + 63) & ~63)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rgb_tmp = torch.empty((B, 3), dtype=torch.float32, device='cuda')
        offs = list(range(0, B, chunk))

        def run_mlp():
            for o in offs:
                n = min(chunk, B - o)
                hlib.check(L.hr_stage_mlp(h, ctypes.c_void_p(rays.data_ptr() + o * rays.shape[1] * 4), n, stream), 'hr_stage_mlp')

        def run_samples():
            for o in offs:
                n = min(chunk, B - o)
                hlib.check(L.hr_stage_samples(h, ctypes.c_void_p(rays.data_ptr() + o * rays.shape[1] * 4), n,
                                              ctypes.c_void_p(rgb_tmp.data_ptr() + o * 12), stream), 'hr_stage_samples')

        # the sample stage reads the head of the LAST chunk the MLP stage wrote: both stages see representative data because
        # every chunk of the frame is statistically alike
        run_mlp(); run_samples()
        reps = max(5, min(args.steps, 20))
        mlp_ms = time_stage(run_mlp, reps)
        smp_ms = time_stage(run_samples, reps)
        nl = len(offs)
        flops = mlp_flops_per_ray(cfg) * B
        byts = algorithmic_bytes_per_ray(cfg, video, texel_bytes) * B
        split = prec_name != 'fp32'
        mlp_kernel = {'bf16x3': 'hr_mlp_bf16x3_kernel', 'f16x3': 'hr_mlp_f16x3_kernel', 'f16x2': 'hr_mlp_f16x2_kernel', 'f16f8': 'hr_mlp_f16f8_kernel', 'fp32': 'hr_mlp_kernel'}[prec_name]
        peak = MFMA_16BIT_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
        n_prod = {'bf16x3': 3, 'f16x3': 3, 'f16x2': 2, 'f16f8': 2, 'fp32': 1}[prec_name]       # (f16f8: one f16 product + one fp8 K=64 instruction per 32 k = two f16 products' pipe time)
        r_mlp = {'kernel': mlp_kernel, 'bound': 'mfma',
                 'achieved': round(flops / (mlp_ms[0] * 1e-3) / 1e12, 3),
                 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(flops / (mlp_ms[0] * 1e-3) / 1e12 / peak, 4),
                 'traffic': None, 'launches_per_step': nl, 'avg_launch_ms': round(mlp_ms[0] / nl, 4),
                 'algorithmic_per_launch': f'{mlp_flops_per_ray(cfg)} FLOP/ray x {min(chunk, B)} rays',
                 'mfma_products_per_gemm': n_prod,
                 'note': (f'fp32 GEMMs evaluated as {n_prod} 16-bit MFMA products of hi/lo split operands, fp32 accumulate: the matrix cores '
                          f'issue {n_prod}x the algorithmic FLOPs, so frac <= 1/{n_prod} by construction.  The datasheet peak needs zero operands: '
                          'with real data the power manager holds the shader clock at 1.84 GHz instead of 2.44 '
                          '(profiles/r02_e_mfma_pipe_ubench.txt), 1.85 PFLOP/s sustained -- frac_of_sustained_issued prices the issued products against that') if split else 'exact fp32 MFMA (v_mfma_f32_16x16x4_f32)'}
        if prec_name == 'f16f8':
            r_mlp['note'] = ('fp32 GEMMs as ONE f16 MFMA product (x_hi w_hi) + one fp8 K=64 MFMA per 32 k for both correction products: the matrix cores issue the pipe time of '
                             '2 f16 products per 16 k, so frac <= 1/2 by construction.  What bounds it (DESIGN.md 3, K1; measured in round 6): the package power limit -- the kernel '
                             'holds the socket at its 1.4 kW cap with the shader clock at 1.91 GHz; without its matrix instructions it runs at 2.38 GHz / 1 270 W, without them and '
                             'the weight loads at 2.39 GHz / 1 150 W (profiles/r06_k1_power_probe.txt).  In cycles the MFMAs overlap the operand stream and the epilogues; in time they '
                             'cost the clock.  The operand stream is not the ceiling (profiles/r06_k1_operand_ubench.txt), and prefetch depth, 4 / 8 wavefronts, staggered or '
                             'turn-taking workgroups all run level (profiles/r06_k1_ab_*.txt): only removing work pays.  frac_of_sustained_issued prices the issued products against '
                             'the 1.85 PFLOP/s the matrix pipe sustains alone under real operands (profiles/r02_e_mfma_pipe_ubench.txt)')
        if split:
            r_mlp['sustained_mfma_tflops'] = 1850.0
            r_mlp['frac_of_sustained_issued'] = round(n_prod * flops / (mlp_ms[0] * 1e-3) / 1e12 / 1850.0, 4)
        gather_tbs = byts / (smp_ms[0] * 1e-3) / 1e12
        r_smp = {'kernel': 'hr_sample_kernel', 'bound': 'valu + vector-L1 load path (co-limited)', 'achieved': None, 'peak': round(VALU_PEAK_GINST, 1),
                 'unit': 'G wave-instructions/s', 'frac': None, 'traffic': None, 'launches_per_step': nl, 'avg_launch_ms': round(smp_ms[0] / nl, 4),
                 'algorithmic_gather_GBs': round(gather_tbs * 1e3, 1),
                 'algorithmic_per_launch': f'{algorithmic_bytes_per_ray(cfg, video, texel_bytes)} B/ray x {min(chunk, B)} rays (L2 / Infinity-Cache resident: not an HBM figure)',
                 'load_path': {'achieved': round(gather_tbs, 2), 'peak': GATHER_UBENCH_L1_TBS, 'peak_l2_resident': GATHER_UBENCH_L2_TBS, 'unit': 'TB/s',
                               'frac': round(gather_tbs / GATHER_UBENCH_L1_TBS, 4),
                               'what': 'the gather\'s algorithmic bytes per second against what the SAME access shape (16-byte lane loads, the 4 lanes of a quad '
                                       'on one 64-byte texel) reaches with nothing else in the kernel: 30.2 TB/s from an L1-resident working set, 16.3 TB/s '
                                       'L2-resident (tools/gather_ubench.hip, profiles/r01_i_gather_ubench.txt); the stage hits L1 ~90 %, L2 ~80 % of the rest'},
                 'note': 'after round 6\'s reductions (1 298 -> ~1 100 vector instructions per sample slot: branch-free density / appearance lanes, one ray '
                         'record per ray, FMA-chain decode sums) the stage is co-limited: VALU ~82 % busy, address unit / L1 path ~73 % -- 13 % fewer '
                         'instructions bought 1.5 % of time, and 5 / 4 / 7 / 8 workgroups per CU instead of 6 are all slower (profiles/r06_k2_*_ab*.txt).  '
                         'VALU side: achieved = VALU wave-instructions per launch (PMC SQ_INSTS_VALU, the committed PMC pass) / live launch time; peak = '
                         '1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (the guide\'s issue rate for plain fp32 instructions; the stage\'s mix of DPP '
                         'and transcendental instructions issues slower: profiles/r03_valu_ilp_ubench.txt).  The library contains no packed-fp32 '
                         'instructions (DESIGN.md 4)'}
        # counters measured separately with rocprofv3 --pmc (never inside a timed run) and committed under profiles/;
        # attached only when the workload matches the profiled one
        tr, state, src = load_counters('', lambda w: (w['model'] == args.model and w['rays_per_launch'] == min(chunk, B) and w['grid'] == grid
                                                      and w['mlp_precision'] == prec_name and w['grid_dtype'] == args.grid_dtype))
        result['counters'] = {'two_kernels': state}
        if tr:
            for r in (r_mlp, r_smp):
                k = tr.get(r['kernel'])
                if not k:
                    continue
                r['traffic'] = k.get('traffic_bytes')
                r['traffic_unit'] = f'HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE, {src})'
                for key in ('limiter', 'mfma_busy_frac', 'lds_conflict_frac', 'l2_hit_rate'):
                    if key in k:
                        r[key] = k[key]
            k = tr.get('hr_sample_kernel')
            if k and 'valu_insts' in k:
                ginst = k['valu_insts'] / (smp_ms[0] / nl * 1e-3) / 1e9
                r_smp['achieved'] = round(ginst, 1)
                r_smp['frac'] = round(ginst / VALU_PEAK_GINST, 4)
                r_smp['valu_insts_per_sample_slot'] = k.get('valu_insts_per_wave')
        result['stage_ms'] = {'mlp': round(mlp_ms[0], 4), 'samples': round(smp_ms[0], 4)}
        if not args.no_extras and not args.no_power:
            # what the power manager does under each kernel and under whole frames (the MLP kernel -- and the frame -- run at the
            # socket's power limit with the shader clock pulled down; the sample stage does not)
            try:
                pw = {'mlp': power_probe(run_mlp), 'samples': power_probe(run_samples), 'frame': power_probe(render_frame)}
                if all(v is not None for v in pw.values()):
                    pw['what'] = ('rocm-smi (socket graphics package power, shader clock; medians over ~1 s each) while the MLP kernel, the sample kernel and '
                                  'whole frames run back to back; outside every timed region')
                    result['power'] = pw
            except Exception as e:       # never the bench line's problem
                result['power'] = {'error': str(e)[:200]}
        if model.frame_kernel_active() and not strong:
            # the step IS one kernel: time its launches with events, price it against the matrix cores (its MLP part is 97 % of
            # the frame's arithmetic) and quote the VALU issue fraction -- what actually limits it -- next to it
            fr_ms = time_stage(lambda: model.render(rays, out=rgb_tmp), reps)
            fk = {'bf16x3': 'hr_frame_bf16x3_kernel', 'f16x3': 'hr_frame_f16x3_kernel', 'f16x2': 'hr_frame_f16x2_kernel', 'f16f8': 'hr_frame_f16f8_kernel'}[prec_name]
            r_fr = {'kernel': fk, 'bound': 'mfma', 'achieved': round(flops / (fr_ms[0] * 1e-3) / 1e12, 3), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(flops / (fr_ms[0] * 1e-3) / 1e12 / peak, 4), 'traffic': None, 'launches_per_step': 1,
                    'avg_launch_ms': round(fr_ms[0], 4), 'algorithmic_per_launch': f'{mlp_flops_per_ray(cfg)} FLOP/ray x {B} rays',
                    'mfma_products_per_gemm': n_prod,
                    'note': 'ONE persistent kernel per frame: MLP wavefronts (matrix cores) and sample wavefronts (vector ALU) of the same '
                            'workgroup, head tile in LDS.  frac prices the whole frame against the 16-bit MFMA peak with the algorithmic MLP FLOPs '
                            f'(<= 1/{n_prod} by construction).  Limits (DESIGN.md 3c-3e): a wavefront issues a DEPENDENT vector instruction only every '
                            '9.3 cycles (tools/valu_ilp_ubench.hip) and the register file holds three wavefronts per SIMD next to the 165-VGPR MLP role '
                            '-- two of them sample wavefronts; and the power manager: under real matrix operands the shader clock is 1.84 GHz against '
                            '2.44 GHz with zero operands (profiles/r02_e_mfma_pipe_ubench.txt).  See valu_busy_frac / mfma_busy_frac',
                    'sustained_mfma_tflops': 1850.0,
                    'frac_of_sustained_issued': round(n_prod * flops / (fr_ms[0] * 1e-3) / 1e12 / 1850.0, 4)}
            r_fr['frac_of_step'] = round(flops / (ms_per_step * 1e-3) / 1e12 / peak, 4)       # the same FLOPs over the timed step (graph replay), not the event-timed launch
            tr, state, src = load_counters('_frame_kernel', lambda w: (w['model'] == args.model and w['rays_per_launch'] == B and w['grid'] == grid
                                                                      and w['mlp_precision'] == prec_name and w['grid_dtype'] == args.grid_dtype))
            result['counters']['frame_kernel'] = state
            if tr and fk in tr:
                k = tr[fk]
                r_fr['traffic'] = k.get('traffic_bytes')
                r_fr['traffic_unit'] = f'HBM-side bytes per launch = per frame (FETCH_SIZE + WRITE_SIZE, {src})'
                for key in ('valu_busy_frac', 'mfma_busy_frac', 'ta_busy_frac', 'lds_conflict_frac', 'limiter'):
                    if key in k:
                        r_fr[key] = k[key]
            result['roofline'] = r_fr
            result['roofline_other'] = {'what': 'the two kernels of the other execution plan (two_kernel_path), timed through hr_stage_*', 'mlp': r_mlp, 'samples': r_smp}
        else:
            dom, oth = (r_mlp, r_smp) if mlp_ms[0] >= smp_ms[0] else (r_smp, r_mlp)
            result['roofline'] = dom
            result['roofline_other'] = oth

    extras = rank == 0 and world == 1 and not args.no_extras and not FAKE and not multi
    # ---- the same frame through the other execution plan, and with the exact fp32-MFMA MLP
    if extras:
        def quick(f):
            g, _ = capture(f.model, rays)
            d = min(timed_frames(g.replay, 20, 5, False, None) for _ in range(2))      # two rounds: a 20-frame window is short enough to catch a noisy neighbour
            return B / (d / 20) / 1e6, d / 20 * 1e3
        verified = bool(model.mlp_verified())
        if verified:
            # the value path is the verified fast path (f16f8 first, the rays at risk again in f16x3: DESIGN 3i).  Beside it, in ONE interleaved
            # window: the round-4 default (f16x3 throughout, two kernels) and plain f16f8 (what the verification costs); and the list's size
            model.render(rays)
            n_redo = model.redo_count()
            result['verified_fast_path'] = {'first_pass': 'f16f8', 'second_pass': 'f16x3 over the rays listed on the device', 'rays_listed': int(n_redo),
                                            'fraction_of_frame': round(n_redo / B, 6), 'list_overflowed': bool(model.redo_overflowed()),
                                            'margins': {k: (round(v, 9) if isinstance(v, float) else v) for k, v in model.verify_info().items()},
                                            'margins_what': 'hr_model_verify_info: per-model margins measured at hr_model_finalize on 4096 synthetic rays (f16f8 vs f16x3 heads through the '
                                                            'model\'s own activations, contraction and intersection); per sample a length is at risk within band * dlen, a distance within '
                                                            'band * dlen * amp, a point within band_q * max amp + band_off (csrc/hr_math.h HrRisk; DESIGN 3c)'}
            base3 = make('f16x3', False)
            v, ms = quick(base3)
            rgb3 = base3.model.render(rays)['rgb'].clone()
            d3 = (rgb3 - rgb).abs().amax(-1)
            result['value_f16x3'] = {'value': round(v, 3), 'unit': 'Mrays/s', 'ms_per_step': round(ms, 4), 'what': 'same frame, f16x3 throughout (the default until round 4), two kernels per chunk',
                                     'linf_vs_value_path': float(d3.max()), 'rays_over_1e-4_vs_value_path': int((d3 > 1e-4).sum())}
            plain8 = make('f16f8', False)
            v8, ms8 = quick(plain8)
            d8 = (plain8.model.render(rays)['rgb'] - rgb3).abs().amax(-1)
            result['value_f16f8_unverified'] = {'value': round(v8, 3), 'unit': 'Mrays/s', 'ms_per_step': round(ms8, 4), 'what': 'same frame, plain f16f8: no list, no second pass',
                                                'rays_over_1e-4_vs_f16x3': int((d8 > 1e-4).sum())}
            if graph is not None:
                g3, _ = capture(base3.model, rays)
                g8, _ = capture(plain8.model, rays)
                result['step_ms_interleaved'] = step_series({'value_path': graph.replay, 'f16x3': g3.replay, 'f16f8_unverified': g8.replay}, 60)
                del g3, g8
            fk = make('f16x3', True)
            if fk.model.frame_kernel_active():
                v, ms = quick(fk)
                result['frame_kernel'] = {'value': round(v, 3), 'unit': 'Mrays/s', 'ms_per_step': round(ms, 4), 'mlp_gemm': 'f16x3',
                                          'bit_identical_to_value_f16x3': bool(torch.equal(fk.model.render(rays)['rgb'], rgb3)),
                                          'what': 'opt-in plan: ONE persistent kernel per frame, MLP wavefronts hand the 64-ray head tile to sample wavefronts of the same workgroup through LDS'}
            del fk, base3, plain8, rgb3
        else:
            other = make(args.mlp_precision, (not use_frame) if use_frame != 2 else False)
            if other.model.frame_kernel_active() != model.frame_kernel_active():
                v, ms = quick(other)
                same = bool(torch.equal(other.model.render(rays)['rgb'], rgb))
                if graph is not None:       # the two plans in ONE window, interleaved replay by replay, an event pair around each
                    g_other, _ = capture(other.model, rays)
                    ab = step_series({'value_path': graph.replay, 'other_plan': g_other.replay}, 100)
                    result['step_ms_interleaved'] = {'value_path': ab['value_path'], 'frame_kernel' if other.model.frame_kernel_active() else 'two_kernel_path': ab['other_plan']}
                    del g_other
                result['frame_kernel' if other.model.frame_kernel_active() else 'two_kernel_path'] = {
                    'value': round(v, 3), 'unit': 'Mrays/s', 'ms_per_step': round(ms, 4), 'bit_identical_to_value_path': same,
                    'what': 'ONE persistent kernel per frame: MLP wavefronts hand the 64-ray head tile to sample wavefronts of the same workgroup '
                            'through LDS (no HBM workspace: 225 MB per 160 000 rays less traffic)' if other.model.frame_kernel_active() else
                            'MLP kernel -> HBM workspace -> sample kernel'}
            del other
        if prec_name != 'fp32':
            exact = make('fp32', False)
            v, ms = quick(exact)
            result['value_fp32_exact'] = {'value': round(v, 3), 'unit': 'Mrays/s', 'ms_per_step': round(ms, 4),
                                          'what': 'same frame, MLP on the exact fp32 MFMA (v_mfma_f32_16x16x4_f32)',
                                          'linf_vs_value_path': float((exact.model.render(rays)['rgb'] - rgb).abs().max())}
            del exact
        if prec_name in ('bf16x3', 'f16x3') or verified:
            # the two-product mode (weights rounded once to half): not the headline -- its error leaves little margin on the keyframe
            # families -- but what the same frame costs with 2/3 of the matrix products, with its own parity below
            fast = make('f16x2', use_frame)
            v, ms = quick(fast)
            fast_rgb = fast.model.render(rays)['rgb'].clone()
            result['value_f16x2'] = {'value': round(v, 3), 'unit': 'Mrays/s', 'ms_per_step': round(ms, 4),
                                     'what': 'same frame, MLP GEMMs as two fp16 MFMA products (activations split, weights rounded once to half): opt-in mlp_precision="f16x2"',
                                     'linf_vs_value_path': float((fast_rgb - rgb).abs().max())}
            del fast
            # the f16 + fp8 mode: the leading product as an f16 MFMA, the two correction products as ONE fp8 (e4m3, block-scaled) K=64 MFMA per
            # 32 k -- the matrix-pipe time of f16x2 at 1/16 of its head error; both execution plans, the faster one reported
            best = None
            for plan in (() if verified else ((False, True) if use_frame else (False,))):
                f8 = make('f16f8', plan)
                v, ms = quick(f8)
                if best is None or ms < best[1]:
                    best = (v, ms, f8.model.render(rays)['rgb'].clone(), bool(f8.model.frame_kernel_active()))
                del f8
            f8_rgb = best[2] if best else None
            if best:
                result['value_f16f8'] = {'value': round(best[0], 3), 'unit': 'Mrays/s', 'ms_per_step': round(best[1], 4),
                                     'what': 'same frame, MLP GEMMs as one f16 MFMA product + one fp8 e4m3 K=64 MFMA for both correction products '
                                             '(v_mfma_scale_f32_32x32x64_f8f6f4): opt-in mlp_precision="f16f8"',
                                     'execution': 'frame kernel' if best[3] else 'two kernels per workspace chunk',
                                     'linf_vs_value_path': float((f8_rgb - rgb).abs().max())}
        if args.grid_dtype == 'fp32':
            # float16 texel STORAGE (the viewer's setting, BASELINE configs[4]): a different (rounded) scene, so not the headline;
            # its parity is held against the reference algorithm on the rounded grids in tests/test_gpu_parity.py
            half = make(args.mlp_precision, use_frame, 'fp16')
            v, ms = quick(half)
            h_rgb = half.model.render(rays)['rgb']
            mse = float(torch.mean((h_rgb - rgb) ** 2))
            result['value_fp16_texels'] = {'value': round(v, 3), 'unit': 'Mrays/s', 'ms_per_step': round(ms, 4),
                                           'what': 'same frame, feature grids stored as float16 (grid_dtype="fp16", class-specialised octet gather), arithmetic in fp32',
                                           'psnr_vs_value_path_db': round(10.0 * float(np.log10(1.0 / max(mse, 1e-20))), 2),
                                           'linf_vs_value_path': float((h_rgb - rgb).abs().max())}
            del half, h_rgb
        torch.cuda.empty_cache()

    # ---- CPU baseline (rank 0, N = 1)
    if rank == 0 and world == 1 and args.cpu_sample > 0 and not FAKE:
        v, secs, idx, ref_rgb = cpu_baseline(cfg, ds, sd_ref, rays_np, min(args.cpu_sample, B))
        got = rgb[torch.from_numpy(idx).cuda()].cpu().numpy()
        sample = (f'{len(idx)} rays of the same frame through oracle/torch_port.py (the reference\'s algorithm on PyTorch CPU ops, fp32, '
                  f'chunk 16384) in {secs:.1f} s')
        try:
            cal = json.load(open(os.path.join(ROOT, 'profiles', 'r02_cpu_calibration.json')))
            r0 = cal['runs'][0]
            sample += (f'; calibration against the reference itself (authoring container, {r0["threads"]} threads, {cal["rays"]} rays of this frame): '
                       f'reference {r0["reference_mrays_s"]:.3f} vs port {r0["port_mrays_s"]:.3f} Mrays/s (port = {r0["port_over_reference"]:.2f}x the reference, '
                       f'L-inf {r0["linf_port_vs_reference"]:.1e})')
        except (OSError, KeyError, ValueError, IndexError):
            pass
        result['cpu_baseline'] = {'value': round(v / 1e6, 5), 'unit': 'Mrays/s', 'cores': torch.get_num_threads(), 'kind': 'port', 'sample': sample}
        err = np.abs(got - ref_rgb).max(-1)
        result['parity_vs_oracle_linf'] = float(err.max())
        result['parity_rays_over_1e-4'] = int((err > 1e-4).sum())
        if 'value_f16x2' in result:
            e2 = np.abs(fast_rgb[torch.from_numpy(idx).cuda()].cpu().numpy() - ref_rgb).max(-1)
            result['value_f16x2']['parity_vs_oracle_linf'] = float(e2.max())
            result['value_f16x2']['parity_rays_over_1e-4'] = int((e2 > 1e-4).sum())
        if 'value_f16f8' in result:
            e3 = np.abs(f8_rgb[torch.from_numpy(idx).cuda()].cpu().numpy() - ref_rgb).max(-1)
            result['value_f16f8']['parity_vs_oracle_linf'] = float(e3.max())
            result['value_f16f8']['parity_rays_over_1e-4'] = int((e3 > 1e-4).sum())

    if strong:
        result['host_us_per_frame'] = round(host_us, 1)
    if FAKE:
        print(json.dumps(result), flush=True) if rank == 0 else None
        dist.destroy_process_group() if multi else None
        return
    result['grid_dtype'] = args.grid_dtype
    result['config']['launch'] = 'eager (Python -> hr_render per frame)' if args.no_graph else \
        ('hipGraph replay of this rank\'s captured render + all-gather enqueued from Python on the side stream' if strong else 'hipGraph replay of one captured frame')
    result['config']['launch'] += f'; {n_pre} uncounted pre-warm steps (~{args.prewarm} s: the idle GPU\'s clock ramp) before the first warm-up step'
    result['dtype'] = {'bf16x3': 'f32 storage/accumulate; GEMM operands bf16x3 split', 'f16x3': 'f32 storage/accumulate; GEMM operands f16x3 split',
                       'f16x2': 'f32 storage/accumulate; GEMM operands f16x2 (weights rounded to half)',
                       'f16f8': 'f32 storage/accumulate; GEMM operands f16 (leading product) + fp8 e4m3 (correction products)', 'fp32': 'f32'}[prec_name]
    result['mlp_gemm'] = {'bf16x3': 'bf16x3 split on MFMA, fp32 accumulate (raw head within 7e-6 of the fp32 chain)',
                          'f16x3': 'f16x3 split on MFMA: 11+11-bit halves, weights pre-scaled by an exact power of two, fp32 accumulate (raw head within 1e-6 of the fp32 chain)',
                          'f16x2': 'f16x2 on MFMA: activations split in two halfs, weights rounded once to half, fp32 accumulate',
                          'f16f8': 'f16 + fp8 on MFMA: x_hi*w_hi as f16, x*w_lo + x_lo*w_hi as one block-scaled fp8 K=64 product, fp32 accumulate (raw head within 2e-5 of the fp32 chain)',
                          'fp32': 'fp32 MFMA'}[prec_name]
    if model.mlp_verified():
        result['mlp_gemm'] += ('; VERIFIED (HR_MLP_F16F8V, what auto resolves to): rays with a comparison inside its per-model, per-sample margin of flipping (verified_fast_path.margins) '
                               'are listed on the device and rendered again with the f16x3 tiles by a second pass inside the same captured frame')
        result['dtype'] += '; second pass f16x3'
    # ---- comparator for the north star's ">= 10x the reference PyTorch single-GPU rays/s": the same algorithm as stock
    #      PyTorch-ROCm ops on this GPU (oracle/torch_port.py on device 'cuda'; the reference itself cannot travel to the GPU
    #      box).  Reported, never part of `value`.
    if extras:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        from torch_port import TorchPort
        tp = TorchPort(cfg, ds, sd_ref, device='cuda')
        best = None
        for ck in (16384, 1048576):          # the reference's shipped ray_chunk and its demo setting
            tp.render(rays, chunk=ck)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                out_t = tp.render(rays, chunk=ck)['rgb']
            torch.cuda.synchronize()
            r = 3 * B / (time.perf_counter() - t1) / 1e6
            if best is None or r > best[0]:
                best = (r, ck)
        result['pytorch_gpu_baseline'] = {
            'value': round(best[0], 3), 'unit': 'Mrays/s', 'chunk': best[1], 'kind': 'port',
            'what': 'the reference algorithm as stock PyTorch-ROCm ops (grid_sample, addmm, sort, cumprod) on the same MI355X',
            'speedup_of_value': round(value / best[0], 1),
            'linf_vs_hip': float((out_t - rgb).abs().max())}
        try:     # the REFERENCE ITSELF, timed once on an MI355X lease next to this port (oracle/refgen/time_reference_gpu.py): the port's calibration
            cal = json.load(open(os.path.join(ROOT, 'profiles', 'r05_gpu_calibration.json')))
            run = max(cal['runs'], key=lambda r: r['reference_mrays_s'])
            result['pytorch_gpu_baseline']['calibration'] = {
                'source': 'profiles/r05_gpu_calibration.json: /root/reference shipped to one MI355X lease, render_chunked over this frame, no cuda->cpu rewrite',
                'reference_mrays_s': round(run['reference_mrays_s'], 3), 'port_mrays_s': round(run['port_mrays_s'], 3), 'chunk': run['chunk'],
                'port_over_reference': round(run['port_over_reference'], 3), 'linf_port_vs_reference': run['linf_port_vs_reference'],
                'linf_hip_vs_reference_full_frame': run['linf_hip_vs_reference'], 'rays_over_1e-4_hip_vs_reference': run['rays_over_1e-4_hip_vs_reference']}
            result['pytorch_gpu_baseline']['speedup_of_value_vs_reference_itself'] = round(value / (best[0] / run['port_over_reference']), 1)
        except (OSError, KeyError, ValueError):
            pass

    if extras and args.model == 'donerf_sphere':
        try:
            result['viewer_path'] = viewer_figures()
        except Exception as e:                                  # never lose the headline line to an extra
            result['viewer_path'] = {'error': repr(e)}
        try:     # BASELINE configs[2..4]: the keyframe families' frame (technicolor: also "1 GPU train+render", see train_step)
            result['families'] = family_figures()
        except Exception as e:
            result['families'] = {'error': repr(e)}
        try:     # SURVEY 8f-4 / BASELINE configs[2]: one optimizer step of the training loop (nlf/__init__.py:634-709), batch 16 384
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from train_bench import train_step_figures
            result['train_step'] = {m: train_step_figures(m, 16384, 15, torch_gpu=False, blas=False) for m in ('donerf_sphere', 'technicolor_z_plane', 'neural_3d_z_plane', 'immersive_sphere')}
        except Exception as e:
            result['train_step'] = {'error': repr(e)}

    if multi:
        dist.barrier()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which a pipe buffers until the process exits -- i.e. AFTER this line, unless it is
        # flushed first.  The JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass
        print(json.dumps(result), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
