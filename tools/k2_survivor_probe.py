"""How many samples of a wavefront survive the reference's app_mask (weight > weight_thresh, tensorf_no_sample.py:201)?

    python tools/k2_survivor_probe.py [model ...]

K2 puts a ray's Z samples into adjacent lanes; the appearance half of the feature gather is needed only for the survivors.  This prints, per
model on the bench frame, the distribution of survivors (and of valid samples) per 64-lane wavefront -- what a compacted appearance pass
would have to serve."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn


def main():
    names = [a for a in sys.argv[1:]] or ['donerf_sphere', 'technicolor_z_plane', 'neural_3d_z_plane', 'immersive_sphere']
    for name in names:
        cfg, ds = C.model_config(name), C.dataset_scalars(name)
        sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
        grid = [int(v) for v in sd['model.color_model.net.gridSize']]
        f = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision='f16x3')
        f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        rays = torch.from_numpy(scenes.benchmark_rays(name, 800, 800, frame=7)).cuda()
        n = 163840
        r = f.model.render(rays[:n], want=('render_weights', 'sigma', 'distances'))
        w = r['render_weights'].cpu().numpy()
        Z = w.shape[1]
        thr = float(f.model._hc.weight_thresh)
        surv = (w > thr)
        ZP = max(8, 1 << (Z - 1).bit_length())
        pad = np.zeros((n, ZP), bool); pad[:, :Z] = surv
        lanes = pad.reshape(-1, 64) if ZP <= 64 else pad.reshape(-1, ZP // 64, 64).reshape(-1, 64)
        c = lanes.sum(1)
        hist = np.bincount(np.minimum((c + 7) // 8, 8), minlength=9)
        print(f'{name}: Z={Z} ZP={ZP} thresh={thr:g} survivors {surv.mean():.3f} of samples; per wavefront mean {c.mean():.1f}  '
              f'==0: {np.mean(c == 0):.3f}  <=16: {np.mean(c <= 16):.3f}  <=32: {np.mean(c <= 32):.3f}  <=48: {np.mean(c <= 48):.3f}  max {c.max()}')
        print('   histogram of ceil(c/8):', hist.tolist())
        sig = r['sigma'].cpu().numpy()
        print(f'   sigma>0 (valid, dense): {np.mean(sig > 0):.3f}')


if __name__ == '__main__':
    main()
