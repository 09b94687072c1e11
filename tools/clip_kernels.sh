#!/bin/bash
# per-kernel durations inside an actual clip fit:  gpurun -- 'bash tools/clip_kernels.sh [frames] [snapshot_interval]'
ROOT=$(pwd); OUT=$ROOT/gpurun_out/clipk; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o r -- python $ROOT/tools/profile_clip.py ${1:-3} ${2:-10} > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(p)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:16]:
    print(f'{r["Name"].split("(")[0][-46:]:46s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:7.1f} us  total {float(r["TotalDurationNs"])/1e6:7.1f} ms')
PY
grep "^total" "$OUT/run.log"
