#!/bin/bash
# same-lease A/B of the opt-in frame kernel (plain f16f8) between library builds: tools/ab_frame_kernel.sh <lib or "product"> ...
for r in 1 2 3; do for l in "$@"; do
  if [ "$l" = product ]; then a=""; else a="--lib $l"; fi
  python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --no-stage-timing --mlp-precision ${PREC:-f16f8} --frame-kernel $a 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('frame kernel, ${PREC:-f16f8}: $l', d['value'], d['ms_per_step'], d.get('execution'))"
done; done
