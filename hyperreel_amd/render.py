"""The reference's render boundary (nlf/rendering.py) backed by the HIP library.

`HipRenderLightfield` is registered under `render_fn_dict['lightfield_hip']` (and, for a
pure drop-in, under the reference's own key 'lightfield'), honours the constructor and
method signatures INRSystem relies on (nlf/__init__.py:355-364, 486-502) and returns the
same `Dict[str, Tensor]`.  `render_chunked` keeps the reference's signature
(nlf/rendering.py:100-150); the native call takes the whole ray list in one go (the
library chunks internally against a fixed workspace), so the Python loop and the per-key
`torch.cat` of the reference disappear for the default case.
"""
from collections import defaultdict

import torch
from torch import nn

from .models import HipLightfieldModel, model_dict  # noqa: F401


class HipRenderLightfield(nn.Module):
    def __init__(self, model, subdivision, cfg, *args, **kwargs):
        super().__init__()
        if subdivision is not None:
            raise NotImplementedError('subdivision schemes are outside the hot-path scope (no shipped model config uses one)')
        self.net_chunk = kwargs['net_chunk'] if 'net_chunk' in kwargs else 32768   # rendering.py:26-29
        self.model = model

    @staticmethod
    def _views(out):
        for k in list(out.keys()):
            out[k] = out[k].view(-1, out[k].shape[-1])                 # _run_multiple, rendering.py:36-43
        return out

    def forward(self, rays, **render_kwargs):
        rays = rays.view(-1, rays.shape[-1])
        return self._views(self.model(rays, render_kwargs))

    def embed(self, rays, **render_kwargs):
        rays = rays.view(-1, rays.shape[-1])
        return self._views(self.model.embed(rays, render_kwargs))

    def forward_multiple(self, rays, **render_kwargs):
        return self.forward(rays, **render_kwargs)


render_fn_dict = {
    'lightfield': HipRenderLightfield,
    'lightfield_hip': HipRenderLightfield,
}


def render_chunked(rays, render_fn, render_kwargs, chunk):
    """Same contract as nlf/rendering.py:100-150 (including `chunk_args` and list-valued
    outputs), usable with any render_fn."""
    B = rays.shape[0]
    results = defaultdict(list)
    chunk_args = getattr(render_kwargs, 'chunk_args', None)
    for i in range(0, B, chunk):
        if chunk_args is None:
            kw = render_kwargs
        else:
            kw = {}
            for k in render_kwargs.keys():
                if k in chunk_args:
                    kw[k] = {j: render_kwargs[k][j][i:i + chunk] for j in render_kwargs[k]}
                else:
                    kw[k] = render_kwargs[k]
        for k, v in render_fn(rays[i:i + chunk], **kw).items():
            results[k] += [v]
    for k, v in results.items():
        if isinstance(v[0], list):
            if 'weights' in k:
                results[k] = v[0]
            else:
                stacked = torch.cat([torch.stack(item, 0) for item in v], 1)
                results[k] = [stacked[idx] for idx in range(stacked.shape[0])]
        else:
            results[k] = torch.cat(v, 0)
    return results


def build_render_fn(model_cfg, dataset=None, system=None, grid_size=None, net_chunk=32768, device='cuda',
                    mlp_precision='auto', grid_dtype='fp32', frame_kernel=False, sample_waves=None, use_occupancy=False, train_deterministic=False, train_fused_mlp=False):
    """What INRSystem.__init__ does for the render path (nlf/__init__.py:350-364):
    model_dict[cfg.type](cfg, system=...) -> render_fn_dict[cfg.render.type](model, None, cfg.render, net_chunk=...)."""
    kwargs = {'system': system, 'mlp_precision': mlp_precision, 'grid_dtype': grid_dtype, 'frame_kernel': frame_kernel,
              'sample_waves': sample_waves, 'use_occupancy': use_occupancy, 'train_deterministic': train_deterministic, 'train_fused_mlp': train_fused_mlp}
    if dataset is not None:
        kwargs['dataset'] = dataset
    if grid_size is not None:
        kwargs['grid_size'] = grid_size
    model = model_dict[model_cfg['type']](model_cfg, **kwargs)
    key = model_cfg['render']['type']
    fn = render_fn_dict[key](model, None, model_cfg['render'], net_chunk=net_chunk)
    return fn.to(device).eval()
