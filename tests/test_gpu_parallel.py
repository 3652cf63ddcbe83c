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
