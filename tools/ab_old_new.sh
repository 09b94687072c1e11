#!/bin/bash
# same-box A/B of two builds of the library (gflow_amd/libgflow_hip_old.so against libgflow_hip.so; same ABI): the bench window's
# kernels under rocprofv3 and 4-frame clip fits, alternating
cp gflow_amd/libgflow_hip.so /tmp/new.so
for r in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp gflow_amd/libgflow_hip_old.so gflow_amd/libgflow_hip.so; else cp /tmp/new.so gflow_amd/libgflow_hip.so; fi
    echo "[$v]"; bash tools/quick_trace.sh ab 2>&1 | grep -E "blend_bwd|blend_fwd" | head -2
    python tools/clip_repeat.py 5 4 2>&1 | tail -1 | cut -c1-110
  done
done
cp /tmp/new.so gflow_amd/libgflow_hip.so
