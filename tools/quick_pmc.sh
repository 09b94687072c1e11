#!/bin/bash
# SQ counters of bench.py's window (last 40 launches of every kernel)   gpurun -- bash tools/quick_pmc.sh tag [ENV=VAL ...]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/qp_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o r -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-pass --no-clip > $OUT/log 2>&1
cd $ROOT
python - $OUT "${FILTER:-blend}" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
path = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gfl::", "")
    if sys.argv[2] not in n: continue
    per[n][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
for k, cs in per.items():
    print(k)
    out = {}
    for c, d in cs.items():
        vals = [d[i] for i in sorted(d)][-40:]
        out[c] = sum(vals) / len(vals)
    for c, v in out.items():
        print("   %-22s %14.0f" % (c, v))
    if "GRBM_GUI_ACTIVE" in out and "SQ_ACTIVE_INST_VALU" in out:
        print("   VALU busy %.3f   kernel us (GRBM/8/2.4GHz) %.1f" % (4 * out["SQ_ACTIVE_INST_VALU"] / (1024 * out["GRBM_GUI_ACTIVE"] / 8), out["GRBM_GUI_ACTIVE"] / 8 / 2400.0))
PY
