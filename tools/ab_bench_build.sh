#!/bin/bash
# A/B of compile-time constants on ONE box, bench window: per-kernel durations (tools/quick_trace.sh) for every CONSTS string, twice
#   gpurun -- bash tools/ab_bench_build.sh "" "-DGFL_FWD_PARTS=8"
for r in 1 2; do
  for c in "$@"; do
    make -C gflow_amd/csrc clean >/dev/null; make -C gflow_amd/csrc CONSTS="$c" -j8 2>&1 | grep -E " error"
    echo "[$c]"; bash tools/quick_trace.sh ab 2>&1 | grep -E "blend_fwd|blend_bwd" | head -2
    python bench.py --no-clip --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   step %.4f ms  fwd %.1f us' % (d['ms_per_step'], 1e3*d['stage_ms']['blend_fwd']))"
  done
done
make -C gflow_amd/csrc clean >/dev/null; make -C gflow_amd/csrc -j8 2>&1 | grep " error"
