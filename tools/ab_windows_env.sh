#!/bin/bash
# A/B of a library environment switch on ONE box, on bench.py's pinned windows (see tools/ab_windows_build.sh):
#   gpurun -- bash tools/ab_windows_env.sh GFL_FWD_SPLIT_MIN 448 256 640          WIN=camera gpurun -- ...
WIN=${WIN:-joint}
VAR=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v python bench.py --only-window $WIN --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
w=d['step_window_clip'] if '$WIN'=='joint' else d['step_window_camera']
f=d['ms_per_step_repeats']
print('[$VAR=%-8s] first-frame %.4f ms (bwd %.1f fwd %.1f us)   $WIN %.4f ms (bwd %.1f fwd %.1f sort %.1f us)' % ('$v', f['median'], 1e3*d['stage_ms']['blend_bwd'], 1e3*d['stage_ms']['blend_fwd'], w['ms_per_step_repeats']['median'], 1e3*w['stage_ms']['blend_bwd'], 1e3*w['stage_ms']['blend_fwd'], 1e3*w['stage_ms']['tile_sort']))"
  done
done
