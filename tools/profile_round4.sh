#!/bin/bash
# Round-4 evidence on the GPU box (-> gpurun_out/r04_*; the summaries are then copied to profiles/):
#   PMC passes of the two execution plans of the headline workload and of the keyframe families' frame kernel, the counters JSON
#   bench.py quotes, the bench line, rocprofv3 kernel-trace stats of the same commands, the VALU issue micro-benchmark.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-a}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
bash tools/pmc.sh ${T}_two --no-frame-kernel > /dev/null 2>&1
python tools/make_counters.py ${T}_two gpurun_out/r04_counters.json donerf_sphere f16x3 fp32 131072 600 600 600 > gpurun_out/r04_${T}_counters_summary.txt
bash tools/pmc.sh ${T}_frame > /dev/null 2>&1
python tools/make_counters.py ${T}_frame gpurun_out/r04_counters_frame_kernel.json donerf_sphere f16x3 fp32 640000 600 600 600 >> gpurun_out/r04_${T}_counters_summary.txt
for m in technicolor_z_plane immersive_sphere; do     # the 32-ray-tile frame kernel (opt-in plan): HBM-side traffic per frame
  bash tools/pmc.sh ${T}_${m} --model $m --frame-mode 2 > /dev/null 2>&1
  python tools/make_counters.py ${T}_${m} gpurun_out/r04_counters_frame_kernel_${m}.json $m f16x3 fp32 640000 0 0 0 >> gpurun_out/r04_${T}_counters_summary.txt
done
# neural_3d (BASELINE configs[3], the slowest family, 64 samples per ray): its default plan is the two kernels
bash tools/pmc.sh ${T}_neural3d --model neural_3d_z_plane > /dev/null 2>&1
python tools/make_counters.py ${T}_neural3d gpurun_out/r04_counters_neural_3d_z_plane.json neural_3d_z_plane f16x3 fp32 65536 823 617 514 >> gpurun_out/r04_${T}_counters_summary.txt
# the opt-in f16 + fp8 arithmetic (two-kernel plan): how busy the matrix pipe is with two thirds of the products
bash tools/pmc.sh ${T}_f16f8 --no-frame-kernel --mlp-precision f16f8 > /dev/null 2>&1
python tools/make_counters.py ${T}_f16f8 gpurun_out/r04_counters_f16f8.json donerf_sphere f16f8 fp32 131072 600 600 600 >> gpurun_out/r04_${T}_counters_summary.txt
for i in 1 2 3 4 5; do for c in two frame neural3d technicolor_z_plane immersive_sphere f16f8; do [ -f gpurun_out/pmc_${T}_${c}_$i.txt ] && cp gpurun_out/pmc_${T}_${c}_$i.txt gpurun_out/r04_${T}_${c}_pmc_pass$i.txt; done; done
cp gpurun_out/r04_counters.json profiles/r04_counters.json      # so that this run's bench line quotes them
cp gpurun_out/r04_counters_frame_kernel.json profiles/r04_counters_frame_kernel.json
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_${T}_bench.json
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o rp -- python $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-stage-timing --no-extras --no-frame-kernel > /tmp/prof.log 2>&1
cd $R
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f gpurun_out/r04_${T}_kernel_stats.csv; done
cd /tmp && rm -rf /tmp/prof2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o rp -- python $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-stage-timing --no-extras > /tmp/prof2.log 2>&1
cd $R
for f in $(find /tmp/prof2 -name "*kernel_stats*.csv"); do cp $f gpurun_out/r04_${T}_kernel_stats_frame_kernel.csv; done
for f in $(find /tmp/prof2 -name "*kernel_trace*.csv"); do head -40 $f > gpurun_out/r04_${T}_kernel_trace_head.csv; done
# one training step kernel by kernel (eager, the library's optimizer)
cd /tmp && rm -rf /tmp/prof3 && HR_OPT=hip timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o train -- python $R/tools/train_graph_probe.py donerf_sphere > /tmp/prof3.log 2>&1
cd $R
python tools/train_kernel_table.py /tmp/prof3/train_results.db > gpurun_out/r04_${T}_train_step_kernels.txt 2>&1
cat gpurun_out/r04_${T}_counters_summary.txt; python -c "
import json; d=json.load(open('gpurun_out/r04_${T}_bench.json'))
for k in ('value','ms_per_step','dtype','stage_ms','two_kernel_path','value_fp32_exact','value_f16x2','value_f16f8','value_fp16_texels','pytorch_gpu_baseline','cpu_baseline','parity_vs_oracle_linf','parity_rays_over_1e-4','viewer_path','families','train_step'): print(k, d.get(k))
print('roofline', d['roofline']); print('other', d['roofline_other'])
"; head -8 gpurun_out/r04_${T}_kernel_stats.csv; head -5 gpurun_out/r04_${T}_kernel_stats_frame_kernel.csv
