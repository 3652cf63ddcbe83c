"""Generates tests/golden/alpha_mask_*.npz by running the REFERENCE's grid-management code (imported in-process through
ref_shim.py): TensorBase.getDenseAlpha / updateAlphaMask (nlf/nets/tensorf_base.py:381-429), TensorVMSplit.shrink
(:1191-1232) and their keyframe-net overrides (nlf/nets/tensorf_dynamic.py:444-536), on a seeded scene whose density is
carved to a sub-box (hyperreel_amd.scenes.carve_density).

    python oracle/refgen/make_alpha_mask.py

TEST INFRASTRUCTURE ONLY.  The weights are regenerated from the recipe by the tests.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402
from hyperreel_amd import config as C  # noqa: E402
from hyperreel_amd import scenes  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'mask')
CASES = [dict(case='alpha_mask_static', model='donerf_sphere', grid=[20, 18, 16], n1=[12, 10, 8], n2=[9, 9, 9], seed=21),
         dict(case='alpha_mask_video', model='technicolor_z_plane', grid=[18, 16, 14], n1=[10, 9, 8], n2=[7, 8, 9], seed=22),
         # a keyframe net on an UNCARVED scene with a threshold inside the range of its alphas: the mask gets holes INSIDE the box it
         # shrinks to, where the density is not zero -- TensorVMKeyframeTime.compute_alpha (tensorf_dynamic.py:618-643) never consults
         # alphaMask, TensorBase.compute_alpha does, and only such a scene tells the two apart in `alpha2`
         dict(case='alpha_mask_video_open', model='technicolor_z_plane', grid=[18, 16, 14], n1=[10, 9, 8], n2=[7, 8, 9], seed=24, thre=0.0062,
              carve=False),
         # the render-path fixture: an UNCARVED scene and a threshold inside the range of its alphas, so that the mask rejects
         # samples that do carry density and the masked image differs from the shipped one
         dict(case='alpha_mask_render', model='donerf_sphere', grid=[20, 18, 16], n1=[12, 10, 8], n2=[9, 9, 9], seed=23, thre=0.013,
              carve=False, render=True)]


def forward_with_mask_enabled(cls):
    """The reference's own TensorVMNoSample.forward with `if self.alphaMask is not None and False:` (tensorf_no_sample.py:171)
    turned back into `if self.alphaMask is not None:` -- compiled from its source text in its own module namespace."""
    import inspect
    import textwrap
    src = textwrap.dedent(inspect.getsource(cls.forward))
    assert 'if self.alphaMask is not None and False:' in src
    src = src.replace('if self.alphaMask is not None and False:', 'if self.alphaMask is not None:')
    ns = {}
    exec(compile(src, '<tensorf_no_sample.forward, mask enabled>', 'exec'), sys.modules[cls.__module__].__dict__, ns)
    return ns['forward']


def main():
    for c in CASES:
        ds = C.dataset_scalars(c['model'])
        model_cfg = C.model_config(c['model'])

        def ov(cfg):
            cfg.color.net.grid_size = ref_shim.to_attr({'start': list(c['grid']), 'end': list(c['grid'])})
            if 'thre' in c:
                cfg.color.net.alpha_mask_thre = c['thre']
        fn = ref_shim.build_reference(ref_shim.load_model_cfg(c['model'], ov), ds)
        sd = scenes.make_state_dict(model_cfg, ds, c['grid'], c['seed'], 'dense', 1.0)
        if c.get('carve', True):
            sd = scenes.carve_density(sd)
        own = dict(fn.state_dict())
        with torch.no_grad():
            for k, v in sd.items():
                if not k.endswith('gridSize'):
                    own[k].copy_(torch.from_numpy(v))
        net = fn.model.color_model.net
        out = {}
        with ref_shim.cpu_mode(), torch.no_grad():
            alpha1, _ = net.getDenseAlpha(tuple(c['n1']))
            out['alpha1'] = alpha1.numpy().astype(np.float32)
            new_aabb = net.updateAlphaMask(tuple(c['n1']))
            out['mask_volume'] = net.alphaMask.alpha_volume.numpy().astype(np.float32)[0, 0]
            out['new_aabb'] = new_aabb.numpy().astype(np.float32)
            if c.get('render'):
                # render with the occupancy test the reference ships disabled (opt-in path of the HIP renderer), before the
                # shrink: the mask's box is the net's box
                rays = scenes.random_rays(384, c['seed'] + 100, False)
                tr = torch.from_numpy(rays)
                out['rays'] = rays
                out['rgb_plain'] = ref_shim.run_reference(fn, tr)['rgb'].numpy().astype(np.float32)
                cls = type(net)
                shipped = cls.forward
                cls.forward = forward_with_mask_enabled(cls)
                try:
                    out['rgb_masked'] = ref_shim.run_reference(fn, tr)['rgb'].numpy().astype(np.float32)
                finally:
                    cls.forward = shipped
                print('masked vs plain render: L-inf', float(np.abs(out['rgb_masked'] - out['rgb_plain']).max()))
            net.shrink(new_aabb)
            out['aabb_after'] = net.aabb.numpy().astype(np.float32)
            out['grid_after'] = net.gridSize.numpy().astype(np.int64)
            for k, v in net.state_dict().items():
                if 'plane' in k or 'line' in k:
                    out['after.' + k] = v.numpy().astype(np.float32)
            alpha2, _ = net.getDenseAlpha(tuple(c['n2']))        # with the mask in place, on the shrunk grid
            out['alpha2'] = alpha2.numpy().astype(np.float32)
        recipe = dict(c, dataset=ds, checksum=scenes.state_dict_checksum(sd))
        np.savez_compressed(os.path.join(OUT, c['case'] + '.npz'), recipe=np.frombuffer(json.dumps(recipe).encode(), dtype=np.uint8), **out)
        print(c['case'], 'alpha1 max', out['alpha1'].max(), 'kept', out['mask_volume'].mean(), 'new_aabb', out['new_aabb'].tolist(),
              'grid', out['grid_after'].tolist(), 'alpha2 >0', (out['alpha2'] > 0).mean())


if __name__ == '__main__':
    main()
