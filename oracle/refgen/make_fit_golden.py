"""A short TRAINING RUN of the reference itself, as a golden (north star: "PSNR within 0.05 dB of reference").

    python oracle/refgen/make_fit_golden.py            ->  tests/golden/fit/<case>.npz

The reference's own modules (imported through ref_shim.py, CPU) are put in train mode and fitted for N_STEPS Adam steps --
the optimizer loop of INRSystem.training_step (nlf/__init__.py:634-709: forward, MSE image loss, backward, Adam) without
its dataset / regulariser plumbing -- from a seeded "student" scene towards the image a seeded "teacher" scene of the same
architecture renders on the same rays.  Fixed full batch (the same 1024 rays every step), the background draw of
tensorf_no_sample.py:236 pinned to "no white background", converged activation / PE schedules (iteration 1e7).  Stored: the
loss of every step, the final eval-mode image and its PSNR against the target, the target, and the recipe (seeds: both
scenes are regenerated from them by scenes.make_state_dict).  tests/test_gpu_train.py runs the same steps through the HIP
training path (forward_train + torch.optim.Adam on the reference-named parameters) and compares.  TEST INFRASTRUCTURE ONLY."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_shim  # noqa: E402
from hyperreel_amd import config as C  # noqa: E402
from hyperreel_amd import scenes  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'fit')
CASES = {
    # case: (model, grid, student seed, teacher seed, rays: (H, W, frame), Adam learning rate)
    # The sphere scene's loss surface has the reference's own steps in it (a sample crossing `dist <= near`, intersect/base.py:194-203,
    # changes the image by a jump): at lr 2e-4 and seeds 21 / 22 (round 3's fixture) two correct fp32 executions of the reference -- all
    # threads vs one thread -- ended 0.18 dB / 30 % of the loss apart, which made the fixture a yardstick of nothing.  Round 4 searched
    # seeds x learning rates with this script's own loop (profiles/r04_fit_fixture_search.txt) for a run the reference reproduces:
    # seeds 41 / 42 at lr 1e-4 -- 17.6 -> 33.8 dB in 200 steps, reference vs itself 0.017 dB / 0.4 % of the loss.
    'donerf_sphere_fit': ('donerf_sphere', [32, 32, 32], 41, 42, (32, 32, 3), 1e-4),
    'technicolor_z_plane_fit': ('technicolor_z_plane', [32, 32, 32], 23, 24, (32, 32, 5), 2e-4),
}
N_STEPS = 200
LR = 2e-4            # (per case: CASES[...][5])


def build(model, grid, dataset, sd):
    def overrides(cfg):
        cfg.color.net.grid_size = ref_shim.to_attr({'start': list(grid), 'end': list(grid)})
    fn = ref_shim.build_reference(ref_shim.load_model_cfg(model, overrides), dataset)
    own = dict(fn.state_dict())
    with torch.no_grad():
        for k, v in sd.items():
            if not k.endswith('gridSize'):
                own[k].copy_(torch.from_numpy(v))
    return fn


def psnr(a, b):
    return float(10.0 * np.log10(1.0 / max(float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)), 1e-20)))


def fit(model, grid, ds, student_sd, rays, target):
    fn = build(model, grid, ds, student_sd)
    first = ref_shim.run_reference(fn, rays)['rgb'].detach().numpy()
    fn.train()
    params = [p for n, p in fn.named_parameters() if 'dummy' not in n]
    opt = torch.optim.Adam(params, lr=LR)
    losses = []
    real_rand = torch.rand
    torch.rand = lambda *a, **k: torch.full((1,), 0.9)          # tensorf_no_sample.py:236: never the white background
    try:
        for step in range(N_STEPS):
            opt.zero_grad(set_to_none=True)
            with ref_shim.cpu_mode():
                rgb = fn(rays)['rgb']
            loss = ((rgb - target) ** 2).mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        torch.rand = real_rand
    fn.eval()
    final = ref_shim.run_reference(fn, rays)['rgb'].detach().numpy()
    return first, final, losses


def main():
    os.makedirs(OUT, exist_ok=True)
    global LR
    for case, (model, grid, s_seed, t_seed, (H, W, frame), lr) in CASES.items():
        LR = lr
        cfg, ds = C.model_config(model), C.dataset_scalars(model)
        rays_np = np.ascontiguousarray(scenes.benchmark_rays(model, H, W, frame=frame), np.float32)
        rays = torch.from_numpy(rays_np)
        teacher = build(model, grid, ds, scenes.make_state_dict(cfg, ds, grid, t_seed, 'dense', 1.0))
        target = ref_shim.run_reference(teacher, rays)['rgb'].detach().clone()
        student_sd = scenes.make_state_dict(cfg, ds, grid, s_seed, 'dense', 1.0)
        threads = torch.get_num_threads()
        first, final, losses = fit(model, grid, ds, student_sd, rays, target)
        # the reference against ITSELF: the same run on one thread (other summation orders inside its GEMMs and reductions) -- how far
        # two correct fp32 executions of this optimisation drift apart; the yardstick for the HIP run's tolerance
        torch.set_num_threads(1)
        _, final_alt, losses_alt = fit(model, grid, ds, student_sd, rays, target)
        torch.set_num_threads(threads)
        recipe = {'case': case, 'model': model, 'grid': grid, 'student_seed': s_seed, 'teacher_seed': t_seed, 'rays': [H, W, frame],
                  'steps': N_STEPS, 'lr': LR, 'dataset': ds, 'student_checksum': scenes.state_dict_checksum(student_sd)}
        path = os.path.join(OUT, case + '.npz')
        np.savez_compressed(path, losses=np.asarray(losses, np.float64), final_rgb=final.astype(np.float32), target=target.numpy().astype(np.float32),
                            psnr_first=np.float64(psnr(first, target.numpy())), psnr_final=np.float64(psnr(final, target.numpy())),
                            psnr_final_alt=np.float64(psnr(final_alt, target.numpy())), losses_alt=np.asarray(losses_alt, np.float64),
                            recipe=np.frombuffer(json.dumps(recipe).encode(), dtype=np.uint8))
        ra = np.abs(np.asarray(losses_alt) - np.asarray(losses)) / np.asarray(losses)
        print(f'{case}: reference vs itself on one thread: PSNR {psnr(final_alt, target.numpy()):.3f} vs {psnr(final, target.numpy()):.3f} dB, '
              f'loss rel max first 40 / all: {ra[:40].max():.2e} / {ra.max():.2e}', flush=True)
        print(f'{case}: loss {losses[0]:.5f} -> {losses[-1]:.5f}, PSNR {psnr(first, target.numpy()):.2f} -> {psnr(final, target.numpy()):.2f} dB, '
              f'{os.path.getsize(path) / 1024:.0f} KiB', flush=True)


if __name__ == '__main__':
    main()
