"""World-size-2 test of the image-parallel path on CPU (gloo): shard -> render -> all-gather.
The local renderer here is the CPU oracle (the tests may use it as the checker); on GPUs the
same `render_sharded` wraps the HIP render_fn over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rays, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, 'oracle')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from hyperreel_amd import config as C
    from hyperreel_amd import scenes
    from hyperreel_amd.parallel import render_sharded, shard_range
    from hyperreel_oracle import HyperReelOracle
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cfg, ds = C.model_config('donerf_sphere', z_channels=16), C.dataset_scalars('donerf')
    sd = scenes.make_state_dict(cfg, ds, [16, 16, 16], seed=2)
    orc = HyperReelOracle(cfg, ds, sd)
    rays = torch.from_numpy(scenes.random_rays(n_rays, 9))
    calls = []

    def local(r):
        calls.append(r.shape[0])
        return torch.from_numpy(orc.render(r.numpy())['rgb'])

    full = render_sharded(local, rays)
    lo, hi = shard_range(n_rays, rank, world)
    assert calls == [hi - lo]
    np.save(os.path.join(out_dir, f'rank{rank}.npy'), full.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, 'ref.npy'), orc.render(rays.numpy())['rgb'])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_rays', [64, 37])
def test_sharded_render_matches_single_process(tmp_path, n_rays):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_rays, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / 'ref.npy')
    for r in range(world):
        got = np.load(tmp_path / f'rank{r}.npy')
        assert got.shape == (n_rays, 3)
        assert np.array_equal(got, ref)       # same oracle, same rays: the gather must not change a bit
