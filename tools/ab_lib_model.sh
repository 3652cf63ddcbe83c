#!/bin/bash
# same-lease A/B of whole-library builds on one model's 800x800 frame: tools/ab_lib_model.sh <model> <lib or "product"> ...   (three interleaved rounds)
m=$1; shift
for r in 1 2 3; do for l in "$@"; do
  if [ "$l" = product ]; then a=""; else a="--lib $l"; fi
  python bench.py --model $m --steps 20 --warmup 5 --no-extras --cpu-sample 0 $a 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m $l', d['value'], d['ms_per_step'], d['stage_ms'])"
done; done
