"""`-m gpu`: the multi-GPU hand-over on the device with a REAL RCCL communicator.  A one-GPU box cannot hold two ranks (RCCL
refuses two ranks on one device), so the group has one rank; what runs is everything but the wire: backend "nccl" (= RCCL on
ROCm), `all_gather_into_tensor` on device tensors, and the strong-scaling pipeline's side stream / events / double buffering
(`ShardedFramePipeline(always_gather=True)`).  The arithmetic of world sizes > 1 is covered on CPU (tests/test_parallel_gloo.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from helpers import Golden

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_group():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', world_size=1, rank=0)
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_sharded_render_and_frame_pipeline_over_rccl(rccl_group):
    from gpu_common import make_render_fn
    from hyperreel_amd import parallel
    g = Golden('donerf_sphere_small')
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict)
    rays = torch.from_numpy(np.ascontiguousarray(g.rays, np.float32)).cuda()
    n = rays.shape[0]
    ref = fn.model.render(rays)['rgb'].clone()
    # ray-sharded render + one all-gather (weak scaling, bench.py's path)
    out = parallel.render_sharded(lambda r: fn.model.render(r)['rgb'], rays)
    assert torch.equal(out, ref)
    # frame sequence: the gather of frame i on the side stream under the render of frame i + 1
    pipe = parallel.ShardedFramePipeline(n, 'cuda', always_gather=True)
    frames = [rays, rays.flip(0).contiguous(), rays.roll(7, 0).contiguous(), rays]
    refs = [fn.model.render(f)['rgb'].clone() for f in frames]
    got = []
    for f in frames:
        tile = pipe.begin()
        fn.model.render(f[pipe.lo:pipe.hi], out=tile)
        prev = pipe.submit()
        if prev is not None:
            got.append(prev.clone())
    got.append(pipe.flush().clone())
    torch.cuda.synchronize()
    assert len(got) == len(frames)
    for a, b in zip(got, refs):
        assert torch.equal(a, b)


def test_frame_pipeline_steps_replayed_from_hipgraphs(rccl_group):
    """strong scaling as bench.py runs it: this rank's render is captured once per tile buffer; a step = graph launch + gather"""
    from gpu_common import make_render_fn
    from hyperreel_amd import parallel
    g = Golden('donerf_sphere_small')
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict)
    rays = torch.from_numpy(np.ascontiguousarray(g.rays, np.float32)).cuda()
    ref = fn.model.render(rays)['rgb'].clone()
    pipe = parallel.ShardedFramePipeline(rays.shape[0], 'cuda', always_gather=True)
    mine = rays[pipe.lo:pipe.hi].contiguous()
    step = pipe.capture(lambda tile: fn.model.render(mine, out=tile))
    got = [f for f in (step() for _ in range(5)) if f is not None] + [pipe.flush()]
    torch.cuda.synchronize()
    assert len(got) == 5 and all(torch.equal(f, ref) for f in got)


def test_c_abi_allgather_of_tiles_on_a_raw_rccl_communicator():
    """hr_allgather_tiles with an ncclComm_t that did NOT come from torch.distributed: a pure-C integrator's path (one rank: a
    one-GPU box cannot hold two)"""
    import ctypes as C
    from gpu_common import make_render_fn
    from hyperreel_amd import lib as hl
    L = hl.load()
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [('internal', C.c_char * 128)]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        g = Golden('donerf_sphere_small')
        fn = make_render_fn(g.cfg, g.dataset, g.state_dict)
        rays = torch.from_numpy(np.ascontiguousarray(g.rays, np.float32)).cuda()
        n = rays.shape[0]
        first, count = C.c_int64(-1), C.c_int64(-1)
        hl.check(L.hr_shard_range(n, 0, 1, C.byref(first), C.byref(count)), 'hr_shard_range')
        assert (first.value, count.value) == (0, n)
        tile = fn.model.render(rays)['rgb'].clone()
        full = torch.full((n, 3), float('nan'), device='cuda')
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        hl.check(L.hr_allgather_tiles(comm, C.c_void_p(tile.data_ptr()), C.c_void_p(full.data_ptr()), 3 * n, stream), 'hr_allgather_tiles')
        torch.cuda.synchronize()
        assert torch.equal(full, tile)
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_bench_strong_scaling_code_path_runs_with_a_forced_one_rank_group():
    """`bench.py --scaling strong` is what measures BASELINE's "800x800 frame ms at 1/2/4/8"; with HR_BENCH_FORCE_DIST=1 its
    multi-rank path (RCCL init, pipeline, barrier, max-reduce) runs on one GPU"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HR_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--scaling', 'strong', '--steps', '6', '--warmup', '2', '--height', '200',
                          '--width', '200', '--no-extras', '--cpu-sample', '0', '--no-stage-timing'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['scaling'] == 'strong' and d['n_gpus'] == 1 and d['value'] > 0
    assert d['host_us_per_frame'] > 0 and 'hipGraph' in d['config']['launch']
