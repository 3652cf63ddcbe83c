"""Image-parallel rendering: one process per GPU, contiguous ray ranges, one all-gather.

Rays are independent on this path (`render_chunked`,
nlf/rendering.py:100-150, already proves order independence), so a frame is split into
`world_size` contiguous ray ranges, every rank renders its range with the replicated scene
and a single `all_gather_into_tensor` of fp32 (rays/N, 3) tiles assembles the frame on every
rank -- over RCCL/xGMI on GPUs (backend "nccl"), over gloo in the CPU tests.  The reference
has no counterpart: it shards whole validation images across DDP ranks and never gathers
pixels (nlf/__init__.py:896).

Training is data-parallel over rays: every rank runs the step on its own batch and `FlatGradients` sums the
gradients with ONE all-reduce of one contiguous buffer (planes + lines + MLP: ~48 MB for the 600^3 DoNeRF scene) --
xGMI is point-to-point, a ring collective is bound per link, so one large message beats the reference's per-bucket
DDP reductions (pytorch_lightning DDP, 25 MB buckets).
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


class ShardedFramePipeline:
    """Strong scaling of a frame SEQUENCE (the viewer / video-render case, BASELINE configs[3-4]): every rank renders its
    contiguous pixel range of frame i into one of two tile buffers, and the all-gather of frame i runs on its own stream
    under the render of frame i + 1 -- over point-to-point xGMI a (B/N, 3) fp32 gather of an 800x800 frame is 7.7 MB in
    total, i.e. tens of microseconds of link time, but its launch + rendezvous latency is of the order of a 1/N-th frame;
    double buffering takes it off the critical path.

        pipe = ShardedFramePipeline(n_pixels, device)
        for frame in frames:
            tile = pipe.begin()                       # (my pixels, 3) buffer to render into (waits until its gather is over)
            model.render(my_rays(frame), out=tile)    # enqueue on the current stream
            done = pipe.submit()                      # all-gather on the side stream; returns the previous frame's image
        last = pipe.flush()

    `render` is the caller's business (HipLightfieldModel.render / render_camera with pixel_range); on CPU (gloo tests) the
    gathers are synchronous.  Every rank ends up with every full frame, in submission order."""

    def __init__(self, n_pixels, device, group=None, channels=3, dtype=torch.float32, always_gather=False):
        self.group = group
        # always_gather: take the collective (and, on a GPU, the side-stream) path even in a one-rank group -- how the RCCL
        # hand-over is exercised on a single-GPU box (tests/test_gpu_parallel.py)
        self.always_gather = bool(always_gather) and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.n = int(n_pixels)
        self.bounds = shard_bounds(self.n, self.world)
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.per = self.bounds[1] - self.bounds[0]                  # largest shard: every rank contributes `per` rows
        dev = torch.device(device)
        self.cuda = dev.type == 'cuda'
        self.tiles = [torch.zeros((self.per, channels), dtype=dtype, device=dev) for _ in range(2)]
        self.full = [torch.empty((self.world * self.per, channels), dtype=dtype, device=dev) for _ in range(2)]
        self.comm = torch.cuda.Stream(dev) if self.cuda else None
        self.rendered = [torch.cuda.Event() if self.cuda else None for _ in range(2)]
        self.pending = [None, None]                                  # async work handle of the gather that reads tiles[slot]
        self.slot = 0
        self.have_prev = False

    def _wait(self, slot):
        w = self.pending[slot]
        if w is not None:
            w.wait()                                                 # the current stream waits for that gather
            self.pending[slot] = None

    def begin(self):
        """The tile buffer of the next frame: rows [0, hi - lo) are this rank's pixels."""
        self._wait(self.slot)                                        # its previous gather (two frames ago) must have read it
        return self.tiles[self.slot][:self.hi - self.lo]

    def _assemble(self, slot):
        """The full frame of `slot`, ALWAYS a fresh tensor: the slot's buffers are written again two submits later, and a caller that
        collects frames (video rendering) must not see earlier ones change under it."""
        out = self.full[slot]
        if self.world * self.per == self.n:
            return out[:self.n].clone()
        return torch.cat([out[r * self.per:r * self.per + (self.bounds[r + 1] - self.bounds[r])] for r in range(self.world)], 0)

    def submit(self):
        """Starts the gather of the frame just rendered into begin()'s buffer; returns the PREVIOUS frame's full image
        (None for the first call) -- valid on the current stream."""
        s = self.slot
        if self.world == 1 and not self.always_gather:
            self.full[s][:self.n].copy_(self.tiles[s][:self.n])
        elif self.cuda:
            self.rendered[s].record()
            self.comm.wait_event(self.rendered[s])
            with torch.cuda.stream(self.comm):
                self.pending[s] = dist.all_gather_into_tensor(self.full[s], self.tiles[s], group=self.group, async_op=True)
        else:
            dist.all_gather_into_tensor(self.full[s], self.tiles[s], group=self.group)
        prev = None
        if self.have_prev:
            self._wait(1 - s)
            prev = self._assemble(1 - s)
        self.have_prev = True
        self.slot = 1 - s
        return prev

    def capture(self, render_into):
        """Captures `render_into(tile)` -- the launches that render this rank's pixels into `tile` -- once per tile buffer as a
        hipGraph, so that a step of the pipeline is ONE graph launch + the all-gather enqueue instead of the Python path through
        hr_render (at 8 ranks a rank's share of an 800x800 frame is ~0.3 ms of kernel: host time per frame matters).  Returns
        `step()`: begin -> replay -> submit.  The rays / camera the launches read must stay where they are between steps."""
        if not self.cuda:
            def step_eager():
                render_into(self.begin())
                return self.submit()
            return step_eager
        graphs = []
        for s in range(2):
            tile = self.tiles[s][:self.hi - self.lo]
            render_into(tile)                                    # warm (allocations, lazy module state) outside the capture
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                render_into(tile)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                render_into(tile)
            graphs.append(g)

        def step():
            self.begin()
            graphs[self.slot].replay()
            return self.submit()
        return step

    def flush(self):
        """The last submitted frame's full image."""
        if not self.have_prev:
            return None
        s = 1 - self.slot
        self._wait(s)
        self.have_prev = False
        return self._assemble(s)


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


class FlatGradients:
    """Gradient storage of a set of parameters as views into ONE flat buffer, so that the data-parallel reduction of a
    training step is a single all-reduce with no packing copies:

        flat = FlatGradients(model)               # the module: follows set_iter's shrink / growth (new Parameters)
        for batch in loader:
            flat.zero()
            loss(model(batch)).backward()        # autograd accumulates into the views
            flat.all_reduce()                    # RCCL over xGMI (backend "nccl"), gloo in the CPU tests
            optimizer.step()

    Use `flat.zero()` instead of `optimizer.zero_grad()` (whose set_to_none would drop the views).
    The reference's counterpart is Lightning's DDP wrapper around INRSystem (bucketed all-reduce inside backward)."""

    def __init__(self, params):
        """params: an nn.Module, a callable returning the parameters, or an iterable of parameters.  With a module or a
        callable the parameter SET is re-read on every zero(): HostTensorVM.set_iter replaces the planes / lines by new
        nn.Parameter objects when it shrinks or grows the grid (tensorf_base.py:510-552), and the flat buffer and its
        views are rebuilt whenever identities or shapes changed.  A plain iterable is a fixed set; zero() then raises if
        a parameter it was given has been detached from its module meanwhile (a stale set would silently stop reducing
        the grid gradients)."""
        if isinstance(params, torch.nn.Module):
            self._source = params.parameters
            self._module = params
        elif callable(params):
            self._source = params
            self._module = None
        else:
            fixed = list(params)
            self._source = lambda: fixed
            self._module = None
        self.params = []
        self.flat = None
        self._rebuild()

    def _current(self):
        return [p for p in self._source() if p.requires_grad]

    def _signature(self, params):
        return [(id(p), tuple(p.shape)) for p in params]

    def _rebuild(self):
        params = self._current()
        if not params:
            raise ValueError('no trainable parameters')
        dev, dt = params[0].device, params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in params):
            raise ValueError('FlatGradients needs all parameters on one device with one dtype')
        self.params = params
        self._sig = self._signature(params)
        self.flat = torch.zeros(sum(p.numel() for p in params), dtype=dt, device=dev)
        self.attach()

    def attach(self):
        """(Re)points every .grad at its slice of the flat buffer (needed again after anything set a .grad to None)."""
        o = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[o:o + n].view(p.shape)
            o += n

    def _inside(self, p):
        lo = self.flat.data_ptr()
        return p.grad is not None and lo <= p.grad.data_ptr() < lo + self.flat.numel() * self.flat.element_size()

    def zero(self):
        if self._signature(self._current()) != self._sig:       # grid shrink / growth replaced parameters
            self._rebuild()
        elif not all(self._inside(p) for p in self.params):
            self.attach()
        self.flat.zero_()

    def check(self, module=None):
        """Raises if a trainable parameter of `module` (default: the module given at construction) has a gradient that
        is not a view of the flat buffer -- such a gradient would be skipped by all_reduce()."""
        module = module if module is not None else self._module
        if module is None:
            return
        for name, p in module.named_parameters():
            if p.requires_grad and not self._inside(p):
                raise RuntimeError(f'FlatGradients: the gradient of {name} is outside the flat buffer (parameter replaced '
                                   f'since the last zero()?); call zero() at the top of every step')

    def all_reduce(self, group=None, average=True):
        """Sum (or mean) of the gradients over the ranks, in place.  A single-process run is a no-op."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        self.check()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            self.flat.div_(world)
