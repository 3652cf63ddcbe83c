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
         dict(case='alpha_mask_video', model='technicolor_z_plane', grid=[18, 16, 14], n1=[10, 9, 8], n2=[7, 8, 9], seed=22)]


def main():
    for c in CASES:
        ds = C.dataset_scalars(c['model'])
        model_cfg = C.model_config(c['model'])

        def ov(cfg):
            cfg.color.net.grid_size = ref_shim.to_attr({'start': list(c['grid']), 'end': list(c['grid'])})
        fn = ref_shim.build_reference(ref_shim.load_model_cfg(c['model'], ov), ds)
        sd = scenes.carve_density(scenes.make_state_dict(model_cfg, ds, c['grid'], c['seed'], 'dense', 1.0))
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
