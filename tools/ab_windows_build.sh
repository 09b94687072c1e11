#!/bin/bash
# A/B of compile-time constants on ONE box, on bench.py's pinned windows: for every CONSTS string, three times over, the
# first-frame window and the mid-clip window named by WIN (default joint): step (median of the repeats) and the blend kernels'
# stage times by the library's events.
#   gpurun -- bash tools/ab_windows_build.sh "" "-DGFL_BWD_LOW_LANES=4"          WIN=camera gpurun -- ...
WIN=${WIN:-joint}
for r in 1 2 3; do
  for c in "$@"; do
    make -C gflow_amd/csrc clean >/dev/null; make -C gflow_amd/csrc CONSTS="$c" -j16 2>&1 | grep -E " error"
    python bench.py --only-window $WIN --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
w=d['step_window_clip'] if '$WIN'=='joint' else d['step_window_camera']
f=d['ms_per_step_repeats']
print('[%-28s] first-frame %.4f ms (bwd %.1f fwd %.1f us)   $WIN %.4f ms (bwd %.1f fwd %.1f us)' % ('$c', f['median'], 1e3*d['stage_ms']['blend_bwd'], 1e3*d['stage_ms']['blend_fwd'], w['ms_per_step_repeats']['median'], 1e3*w['stage_ms']['blend_bwd'], 1e3*w['stage_ms']['blend_fwd']))"
  done
done
make -C gflow_amd/csrc clean >/dev/null; make -C gflow_amd/csrc -j16 2>&1 | grep " error"
