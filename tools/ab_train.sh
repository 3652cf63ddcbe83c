#!/bin/bash
# same-lease A/B of the training step between library builds: tools/ab_train.sh <lib or "product"> ...   (two interleaved rounds, four models)
for r in 1 2; do for m in donerf_sphere technicolor_z_plane neural_3d_z_plane immersive_sphere; do for l in "$@"; do
  if [ "$l" = product ]; then e="env"; else e="env HR_LIB=$l"; fi
  $e python tools/train_bench.py --model $m --steps 30 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m $l', {k: d[k] for k in ('hip_ms_per_step','hip_ms_forward_backward','hip_ms_sample_stage_backward','hip_ms_mlp_forward_backward') if k in d})"
done; done; done
