#!/bin/bash
# host-side HIP API time inside a clip fit:  gpurun -- 'bash tools/clip_hip_api.sh [frames] [snapshot_interval]'
ROOT=$(pwd); OUT=$ROOT/gpurun_out/cliphip; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d "$OUT" -o r -- python $ROOT/tools/profile_clip.py ${1:-3} ${2:-10} > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*hip_api_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(p)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print(f'{r["Name"]:40s} calls {int(r["Calls"]):7d}  avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.1f} ms  max {float(r["MaxNs"])/1e6:7.2f} ms')
PY
grep "^total\|^train\|^dens\|^make" "$OUT/run.log"
