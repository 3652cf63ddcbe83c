#!/usr/bin/env python
"""Runs bench.py over a list of option sets and prints one summary line per run.
usage: python tools/sweep.py "--chunk 32768" "--chunk 65536 --mlp-precision fp32" ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for opts in sys.argv[1:]:
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '100', '--warmup', '20', '--cpu-sample', '0'] + opts.split()
    p = subprocess.run(cmd, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith('{')]
    if not line:
        print(opts, 'FAILED', p.stderr[-400:])
        continue
    d = json.loads(line[-1])
    print(f"{opts:45s} {d['value']:8.2f} Mrays/s  {d['ms_per_step']:.3f} ms/frame  stages {d.get('stage_ms')}", flush=True)
