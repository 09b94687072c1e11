#!/bin/bash
# A/B of an environment switch on the clip fit only, alternating, ONE box:  gpurun -- 'bash tools/ab_env_clip.sh VAR=1 [rounds]'
SW=$1
for r in $(seq 1 ${2:-4}); do
  for v in A B; do
    if [ $v = B ]; then export $SW; else unset ${SW%%=*}; fi
    echo -n "$v: clip "; python tools/profile_clip.py 8 10 | grep "^total" | cut -d= -f2
  done
done
