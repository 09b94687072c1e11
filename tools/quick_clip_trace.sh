#!/bin/bash
# per-kernel average durations inside a 3-frame clip fit under rocprofv3   gpurun -- bash tools/quick_clip_trace.sh tag [ENV=VAL ...]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/qc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $ROOT/tools/profile_clip.py 3 10 > $OUT/log 2>&1
cd $ROOT
python - $OUT <<'PY'
import csv, glob, sys, os
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(path)):
    n = r["Name"].split("(")[0].replace("void ", "").replace("gfl::", "")
    if "at::" in n or float(r["Percentage"]) < 0.8: continue
    print("%8.1f us  x%-6s %5.1f%%  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["Percentage"]), n[:90]))
PY
