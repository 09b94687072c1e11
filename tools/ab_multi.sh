#!/bin/bash
# A/B/C... of several builds of the library on ONE box: copy them to tmp_ab/lib<NAME>.so, then
#   gpurun -- bash tools/ab_multi.sh [rounds] NAME1 NAME2 ...
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp tmp_ab/lib$v.so gflow_amd/libgflow_hip.so
    echo -n "$v: clip "; python tools/profile_clip.py 8 10 | grep "^total" | cut -d= -f2
    echo -n "$v: step "; python bench.py --steps 200 --warmup 50 --no-clip --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in s.items()})"
  done
done
