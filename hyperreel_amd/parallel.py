"""Image-parallel rendering: one process per GPU, contiguous ray ranges, one all-gather.

Rays are independent on this path (`render_chunked`,
nlf/rendering.py:100-150, already proves order independence), so a frame is split into
`world_size` contiguous ray ranges, every rank renders its range with the replicated scene
and a single `all_gather_into_tensor` of fp32 (rays/N, 3) tiles assembles the frame on every
rank -- over RCCL/xGMI on GPUs (backend "nccl"), over gloo in the CPU tests.  The reference
has no counterpart: it shards whole validation images across DDP ranks and never gathers
pixels (nlf/__init__.py:896).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rays, world_size):
    """Even contiguous partition: the first `n_rays % world_size` ranks take one extra ray."""
    base, extra = divmod(int(n_rays), int(world_size))
    bounds = [0]
    for r in range(world_size):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


def shard_range(n_rays, rank, world_size):
    b = shard_bounds(n_rays, world_size)
    return b[rank], b[rank + 1]


def render_sharded(render_local, rays, group=None):
    """rays: the FULL (B, C) ray list, identical on every rank.  render_local(rays_slice) ->
    (n, 3) tensor on the same device.  Returns the full (B, 3) image on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return render_local(rays)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = rays.shape[0]
    bounds = shard_bounds(B, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    tile = render_local(rays[lo:hi])
    per = bounds[1] - bounds[0]                       # largest shard (rank 0 always has it)
    if tile.shape[0] < per:                           # pad so that every rank contributes `per` rows
        tile = torch.cat([tile, tile.new_zeros((per - tile.shape[0], tile.shape[1]))], 0)
    out = tile.new_empty((world * per, tile.shape[1]))
    dist.all_gather_into_tensor(out, tile.contiguous(), group=group)
    if world * per == B:
        return out
    pieces = [out[r * per:r * per + (bounds[r + 1] - bounds[r])] for r in range(world)]
    return torch.cat(pieces, 0)


def render_camera_sharded(model, pose, K, width, height, time=None, cam_id=0.0, group=None):
    """Image-parallel frame straight from the camera: every rank generates only the rays of its
    own pixel range on its GPU (`HipLightfieldModel.generate_rays`: 80 bytes of camera instead of a
    scattered ray list), renders them and takes part in one all-gather of the tiles."""
    n = int(width) * int(height)
    if not (dist.is_available() and dist.is_initialized()):
        return model.render_camera(pose, K, width, height, time, cam_id)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds(n, world)
    tile = model.render_camera(pose, K, width, height, time, cam_id, pixel_range=(bounds[rank], bounds[rank + 1]))
    per = bounds[1] - bounds[0]
    if tile.shape[0] < per:
        tile = torch.cat([tile, tile.new_zeros((per - tile.shape[0], 3))], 0)
    out = tile.new_empty((world * per, 3))
    dist.all_gather_into_tensor(out, tile.contiguous(), group=group)
    if world * per == n:
        return out
    return torch.cat([out[r * per:r * per + (bounds[r + 1] - bounds[r])] for r in range(world)], 0)


class ShardedRenderFn(torch.nn.Module):
    """Wraps a render_fn (e.g. HipRenderLightfield) so that `forward(rays)['rgb']` renders
    image-parallel across the default process group."""

    def __init__(self, render_fn, group=None):
        super().__init__()
        self.render_fn = render_fn
        self.group = group

    def forward(self, rays, **render_kwargs):
        rgb = render_sharded(lambda r: self.render_fn(r, **render_kwargs)['rgb'], rays, self.group)
        return {'rgb': rgb}
