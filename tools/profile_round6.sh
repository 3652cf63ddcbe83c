#!/bin/bash
# Round-6 evidence on the GPU box (-> gpurun_out/r06_*; the summaries are then copied to profiles/):
#   PMC passes of the two kernels of the default plan (plain f16f8: the verified path's first pass without its list-driven launches, so that
#   per-dispatch averages are per 160 000-ray launch), the counters JSON bench.py quotes, rocprofv3 kernel-trace stats of the driver's command,
#   the bench line itself (driver command, three times), Neural-3D's counters.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-a}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
P="--prewarm 0 --windows 1"
bash tools/pmc.sh ${T}_f16f8 $P --mlp-precision f16f8 > /dev/null 2>&1
python tools/make_counters.py ${T}_f16f8 gpurun_out/r06_counters.json donerf_sphere f16f8 fp32 160000 600 600 600 > gpurun_out/r06_${T}_counters_summary.txt
bash tools/pmc.sh ${T}_neural3d $P --mlp-precision f16f8 --model neural_3d_z_plane > /dev/null 2>&1
python tools/make_counters.py ${T}_neural3d gpurun_out/r06_counters_neural_3d_z_plane.json neural_3d_z_plane f16f8 fp32 64000 823 617 514 >> gpurun_out/r06_${T}_counters_summary.txt
for i in 1 2 3 4 5; do for c in f16f8 neural3d; do [ -f gpurun_out/pmc_${T}_${c}_$i.txt ] && cp gpurun_out/pmc_${T}_${c}_$i.txt gpurun_out/r06_${T}_${c}_pmc_pass$i.txt; done; done
cp gpurun_out/r06_counters.json profiles/r06_counters.json      # so that this run's bench line quotes them
for i in 1 2 3; do timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_${T}_bench_$i.json; done
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o rp -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-stage-timing --no-extras --prewarm 0.1 --windows 1 > /tmp/prof.log 2>&1
cd $R
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f gpurun_out/r06_${T}_kernel_stats.csv; done
for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do head -60 $f > gpurun_out/r06_${T}_kernel_trace_head.csv; done
cat gpurun_out/r06_${T}_counters_summary.txt; python3 -c "
import json
for i in (1,2,3):
    d=json.load(open('gpurun_out/r06_${T}_bench_%d.json' % i))
    print(i, d['value'], d['ms_per_step'], d['windows_ms_per_step'], d.get('counters'))
d=json.load(open('gpurun_out/r06_${T}_bench_1.json'))
for k in ('dtype','stage_ms','verified_fast_path','value_f16x3','value_f16f8_unverified','value_fp32_exact','value_f16x2','value_fp16_texels','pytorch_gpu_baseline','cpu_baseline','parity_vs_oracle_linf','parity_rays_over_1e-4','viewer_path','families','step_ms','step_ms_interleaved'): print(k, d.get(k))
print('roofline', d['roofline']); print('other', d['roofline_other'])
"; head -12 gpurun_out/r06_${T}_kernel_stats.csv
