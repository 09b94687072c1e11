#!/bin/bash
# A/B of a library environment switch on whole clip fits (the metric's recipe: snapshots, trajectories), alternating, ONE box:
#   gpurun -- bash tools/ab_env_clipfit.sh GFL_FWD_SPLIT_MIN 30 448 352 [rounds]      (frames, then the values)
VAR=$1; FR=$2; shift; shift
for r in 1 2 3; do
  for v in "$@"; do
    echo -n "[$VAR=$v] "; env $VAR=$v python tools/trace_clip.py $FR 10 100 | grep "fit of"
  done
done
