"""`python bench.py --gpus N` must start by itself: with no launcher around it the script re-executes itself under
torch.distributed.run, one process per rank, rendezvous on 127.0.0.1.  Run here at N = 2 on CPU over gloo with bench.py's stand-in
renderer (HR_BENCH_FAKE_RENDERER=1: the launch, rank, window, gather and JSON plumbing is the real one; no figure of such a run means
anything).  Also: the frame the strong window assembles from two ranks' tiles is the single-process image."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ, HR_BENCH_FAKE_RENDERER='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--height', '24', '--width', '20',
                        '--prewarm', '0', '--windows', '2', *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]              # ONE line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_launches_itself_and_prints_one_line_with_the_strong_window():
    d = _run(['--gpus', '2'])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['value'] > 0 and d['ms_per_step'] > 0 and len(d['windows_ms_per_step']) == 2
    assert d['config']['rays_per_gpu'] == 480 and 'x2' in d['config']['parallelism']
    s = d['strong']
    assert s['ranks'] == 2 and s['rccl_ranks'] == 2 and s['rays_per_rank'] == 240
    for k in ('frame_ms', 'mrays_s', 'n1_frame_ms', 'speedup_vs_n1_frame_ms', 'tile_render_ms', 'gather_alone_ms', 'host_us_per_frame'):
        assert s[k] > 0, k
    assert 0.0 <= s['gather_overlap'] <= 1.0
    assert abs(s['speedup_vs_n1_frame_ms'] - s['n1_frame_ms'] / s['frame_ms']) <= 2e-3 * max(1.0, s['speedup_vs_n1_frame_ms'])
    assert 'HR_BENCH_FAKE_RENDERER' in d['data']


def test_strong_scaling_value_at_3_ranks():
    """--scaling strong: `value` is the ONE-frame window (frame rays / frame time), uneven split (480 rays over 3 ranks)."""
    d = _run(['--gpus', '3', '--scaling', 'strong'])
    assert d['n_gpus'] == 3 and d['scaling'] == 'strong'
    assert d['strong']['rays_per_rank'] == 160 and d['config']['rays_per_gpu'] == 160
    assert abs(d['value'] - d['strong']['mrays_s']) <= 1e-2 * d['value'] + 1e-3
    assert d['host_us_per_frame'] > 0


def test_one_rank_takes_the_distributed_path_when_forced():
    # exactly `HR_BENCH_FORCE_DIST=1 python bench.py --gpus 1`: no launcher, no rendezvous variables (bench.py makes itself a one-rank job)
    d = _run(['--gpus', '1'], {'HR_BENCH_FORCE_DIST': '1'})
    assert d['n_gpus'] == 1 and d['strong']['ranks'] == 1 and d['strong']['rccl_ranks'] == 1


def test_a_mismatched_launch_is_refused():
    env = dict(os.environ, HR_BENCH_FAKE_RENDERER='1', WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
