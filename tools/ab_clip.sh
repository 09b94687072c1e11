#!/bin/bash
# A/B of one environment switch on ONE box on the clip fit (4 frames x 6 fits, min / median) and on bench.py's step
#   gpurun -- bash tools/ab_clip.sh GFL_RESERVED 1 0 [repeats]
VAR=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq 1 $R); do
  for v in $A $B; do
    echo -n "$VAR=$v  "; env $VAR=$v python tools/clip_repeat.py 6 4 2>&1 | tail -1 | cut -c1-140
  done
done
for v in $A $B; do
  env $VAR=$v python bench.py --no-clip --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); st=d['stage_ms']; print('$VAR=$v bench step %.4f ms  fwd %.1f bwd %.1f us' % (d['ms_per_step'], 1e3*st['blend_fwd'], 1e3*st['blend_bwd']))"
done
