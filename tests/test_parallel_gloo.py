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


class _FakeCameraModel:
    """Stands in for HipLightfieldModel.render_camera on CPU: colour = a function of the pixel index."""

    def __init__(self):
        self.calls = []

    def render_camera(self, pose, K, width, height, time=None, cam_id=0.0, pixel_range=None):
        lo, hi = (0, width * height) if pixel_range is None else pixel_range
        self.calls.append((lo, hi))
        i = torch.arange(lo, hi, dtype=torch.float32)
        return torch.stack([i, i * 0.5, i % 7], -1)


def _camera_worker(rank, world, port, width, height, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from hyperreel_amd.parallel import render_camera_sharded, shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    m = _FakeCameraModel()
    full = render_camera_sharded(m, None, None, width, height)
    assert m.calls == [shard_range(width * height, rank, world)]      # every rank renders its own pixel range only
    np.save(os.path.join(out_dir, f'cam{rank}.npy'), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('wh', [(8, 6), (7, 5)])
def test_camera_sharding_assembles_the_frame(tmp_path, wh):
    world = 2
    mp.spawn(_camera_worker, args=(world, _free_port(), wh[0], wh[1], str(tmp_path)), nprocs=world, join=True)
    ref = _FakeCameraModel().render_camera(None, None, wh[0], wh[1]).numpy()
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'cam{r}.npy'), ref)


def _train_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from hyperreel_amd.parallel import FlatGradients
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)                                         # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.LeakyReLU(0.01), torch.nn.Linear(16, 3))
    plane = torch.nn.Parameter(torch.rand(1, 4, 8, 8))
    params = list(net.parameters()) + [plane]
    flat = FlatGradients(params)
    opt = torch.optim.Adam(params, lr=1e-2)
    g = torch.Generator().manual_seed(100 + rank)               # every rank its own ray batch
    for step in range(3):
        x = torch.randn(32, 6, generator=g)
        flat.zero()
        loss = (net(x) * plane.mean()).pow(2).mean()
        loss.backward()
        local = flat.flat.clone()
        flat.all_reduce()
        if step == 0:
            torch.save({'local': local, 'reduced': flat.flat.clone()}, os.path.join(out_dir, f'g{rank}.pt'))
        if step == 1:
            opt.zero_grad(set_to_none=True)                      # someone drops the views: zero() must re-attach
            flat.zero()
            (net(x) * plane.mean()).pow(2).mean().backward()
            flat.all_reduce()
        opt.step()
    torch.save([p.detach().clone() for p in params], os.path.join(out_dir, f'p{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradients_all_reduce_keeps_replicas_identical(tmp_path):
    """Data-parallel training step: one all-reduce of the flat gradient buffer, world size 2 on gloo."""
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = [torch.load(tmp_path / f'g{r}.pt') for r in range(world)]
    want = (g[0]['local'] + g[1]['local']) / 2
    assert float(want.abs().max()) > 0 and not torch.equal(g[0]['local'], g[1]['local'])
    for r in range(world):
        assert torch.allclose(g[r]['reduced'], want, rtol=0, atol=1e-7)
    p = [torch.load(tmp_path / f'p{r}.pt') for r in range(world)]
    for a, b in zip(*p):
        assert torch.equal(a, b)                                 # same averaged gradients -> bitwise identical replicas


def _pipeline_worker(rank, world, port, n_pixels, n_frames, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from hyperreel_amd.parallel import ShardedFramePipeline
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    pipe = ShardedFramePipeline(n_pixels, 'cpu')
    frames = []
    for f in range(n_frames):
        tile = pipe.begin()
        i = torch.arange(pipe.lo, pipe.hi, dtype=torch.float32)
        tile.copy_(torch.stack([i + 1000.0 * f, i * 0.5, (i + f) % 7], -1))       # "render": a function of (pixel, frame)
        prev = pipe.submit()
        if prev is not None:
            frames.append(prev.clone())
    frames.append(pipe.flush().clone())
    np.save(os.path.join(out_dir, f'rank{rank}.npy'), torch.stack(frames).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_pixels', [(2, 64), (4, 37), (4, 640)])
def test_double_buffered_frame_pipeline(tmp_path, world, n_pixels):
    """Strong-scaling path of bench.py --scaling strong: ragged pixel split over 4 ranks, frames come back complete, in order,
    on every rank, while the next frame is already being written into the other buffer."""
    n_frames = 5
    mp.spawn(_pipeline_worker, args=(world, _free_port(), n_pixels, n_frames, str(tmp_path)), nprocs=world, join=True)
    i = np.arange(n_pixels, dtype=np.float32)
    ref = np.stack([np.stack([i + 1000.0 * f, i * 0.5, (i + f) % 7], -1) for f in range(n_frames)])
    for r in range(world):
        got = np.load(tmp_path / f'rank{r}.npy')
        assert got.shape == ref.shape and np.array_equal(got, ref), r


def test_flat_gradients_follow_replaced_parameters():
    """HostTensorVM.set_iter replaces planes / lines by new nn.Parameters when it shrinks or grows the grid; FlatGradients built
    from the MODULE must pick the new ones up at the next zero() (all .grad views of one buffer again), and one built from a
    fixed parameter list must refuse to reduce a stale set."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from hyperreel_amd.parallel import FlatGradients

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.plane = torch.nn.Parameter(torch.ones(1, 4, 6, 6))
            self.lin = torch.nn.Linear(3, 2)

    m = Net()
    flat = FlatGradients(m)
    flat.zero()
    (m.plane.sum() + m.lin.weight.sum()).backward()
    assert float(flat.flat.sum()) == 4 * 36 + 6
    n0 = flat.flat.numel()
    m.plane = torch.nn.Parameter(torch.ones(1, 4, 12, 12))           # what upsample_volume_grid does
    flat.zero()
    assert flat.flat.numel() == n0 - 4 * 36 + 4 * 144
    (m.plane.sum() * 2.0).backward()
    flat.check()
    lo = flat.flat.data_ptr()
    assert lo <= m.plane.grad.data_ptr() < lo + flat.flat.numel() * 4 and float(flat.flat.sum()) == 2.0 * 4 * 144
    # a gradient that escaped the buffer is refused
    m.lin.weight.grad = torch.zeros_like(m.lin.weight)
    with pytest.raises(RuntimeError):
        flat.check()
