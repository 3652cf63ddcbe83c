#!/bin/bash
# Round-end evidence on the GPU box: default bench (with CPU baseline + PyTorch-ROCm comparator), the float16-texel
# viewer mode, the other model families, and a rocprofv3 kernel trace of the default command.  -> gpurun_out/g_*
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 600 python bench.py --torch-gpu 2>&1 | tail -1 > gpurun_out/g_bench.json
timeout 300 python bench.py --cpu-sample 0 --mlp-precision fp32 2>&1 | tail -1 > gpurun_out/g_bench_fp32.json
timeout 300 python bench.py --cpu-sample 20000 --grid-dtype fp16 2>&1 | tail -1 > gpurun_out/g_bench_fp16grids.json
for m in donerf_cylinder technicolor_z_plane neural_3d_z_plane immersive_sphere; do
  timeout 300 python bench.py --cpu-sample 20000 --model $m --steps 50 2>&1 | tail -1 > gpurun_out/g_bench_$m.json
done
timeout 300 python bench.py --cpu-sample 0 --model immersive_sphere --grid-dtype fp16 --steps 50 2>&1 | tail -1 > gpurun_out/g_bench_immersive_sphere_fp16grids.json
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o rp -- python $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-stage-timing 2>&1 | grep -v simple_timer | tail -3
cd $R
for f in $(find /tmp/prof -name "*stats*.csv"); do cp $f gpurun_out/prof/g_$(basename $f); done
for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do head -120 $f > gpurun_out/prof/g_$(basename $f); done
cat gpurun_out/g_bench.json; cat gpurun_out/prof/g_rp_kernel_stats.csv | head -12
for f in gpurun_out/g_bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d.get('stage_ms'), d.get('parity_vs_oracle_linf'))"; done
