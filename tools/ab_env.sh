#!/bin/bash
# A/B of an environment switch of the library on ONE box:  gpurun -- 'bash tools/ab_env.sh GFL_CAMERA_KERNEL=1 [rounds]'
# A = default, B = with the switch.  Prints ms_per_step and the stage times of both, alternating.
SW=$1
for r in $(seq 1 ${2:-3}); do
  for v in A B; do
    if [ $v = B ]; then export $SW; else unset ${SW%%=*}; fi
    echo -n "$v: clip "; python tools/profile_clip.py 8 10 | grep "^total" | cut -d= -f2
    echo -n "$v: step "; python bench.py --steps 200 --warmup 50 --no-clip --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in s.items()})"
  done
done
