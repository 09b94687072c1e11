#!/bin/bash
# per-kernel durations of bench.py's timed window under rocprofv3 (last 40 launches of every kernel)
#   gpurun -- bash tools/quick_trace.sh tag [ENV=VAL ...]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/qt_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-pass --no-clip > $OUT/log 2>&1
cd $ROOT
python - $OUT <<'PY'
import csv, glob, sys, os
from collections import defaultdict
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
per = defaultdict(list)
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gfl::", "")
    per[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows = []
for k, v in per.items():
    if "at::" in k: continue
    v.sort(); w = v[-40:]
    rows.append((sum(e - b for b, e in w) / len(w) / 1e3, len(v), k))
for a, n, k in sorted(rows, reverse=True):
    print("%8.1f us  x%-4d %s" % (a, n, k[:110]))
PY
