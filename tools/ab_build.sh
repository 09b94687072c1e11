#!/bin/bash
# A/B of compile-time constants on ONE box: clip fit (6 x 4 frames, median) for every CONSTS string given, twice
#   gpurun -- bash tools/ab_build.sh "" "-DGFL_HEAVY_SEG=128" "-DGFL_HEAVY_SEG=224"
for r in 1 2; do
  for c in "$@"; do
    make -C gflow_amd/csrc clean >/dev/null; make -C gflow_amd/csrc CONSTS="$c" -j8 2>&1 | grep -E " error"
    echo -n "[$c]  "; python tools/clip_repeat.py 6 4 2>&1 | tail -1 | cut -c30-105
  done
done
make -C gflow_amd/csrc clean >/dev/null; make -C gflow_amd/csrc -j8 2>&1 | grep " error"
