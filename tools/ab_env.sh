#!/bin/bash
# same-lease A/B of one environment switch on the headline frame: tools/ab_env.sh VAR=value [bench args...]   (three interleaved rounds, with / without)
kv=$1; shift
for r in 1 2 3; do for on in 1 0; do
  if [ $on = 1 ]; then pre="env $kv"; tag="$kv"; else pre="env"; tag="(unset)"; fi
  $pre python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d.get('stage_ms'), 'parity', d.get('parity_vs_oracle_linf'), d.get('parity_rays_over_1e-4'))"
done; done
