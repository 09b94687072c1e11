#!/bin/bash
# A/B of one environment switch on ONE box, alternating: bench.py's step (pinned window) with VAR=A and VAR=B, R times each.
#   gpurun -- bash tools/ab_env.sh GFL_RESERVED 1 0 [repeats] [extra bench flags]
VAR=$1; A=$2; B=$3; R=${4:-3}; shift 4
for r in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-clip --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
st=d['stage_ms']
print('$VAR=$v', 'ms_per_step %.4f' % d['ms_per_step'], 'K %.0f' % d['config']['splat_tile_pairs_K'], ' '.join('%s %.1f' % (k, 1e3*x) for k,x in st.items()))
"
  done
done
