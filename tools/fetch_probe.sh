#!/bin/bash
# FETCH_SIZE of three kernels that move a known number of bytes (tools/fetch_probe.hip).   gpurun -- bash tools/fetch_probe.sh
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/fetch_probe
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o r -- $ROOT/tools/fetch_probe.bin > "$OUT/probe.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- $ROOT/tools/fetch_probe.bin >> "$OUT/probe.log" 2>&1
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys, json
from collections import defaultdict
root = sys.argv[1]
per = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(glob.glob(os.path.join(root, "fetch", "**", "*counter_collection.csv"), recursive=True)[0])):
    if r["Counter_Name"] == "FETCH_SIZE":
        per[r["Kernel_Name"].split("(")[0]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
dur = {}
for r in csv.DictReader(open(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)[0])):
    dur[r["Name"].split("(")[0]] = float(r["AverageNs"]) / 1e3
K, rows_small = 4 << 20, 60000
known = {"stream_kernel": ("256 MiB streamed once", 256 << 20),
         "gather_kernel": ("ids 16 MiB + the 2.9 MB table (once per XCD at most: 23 MB)", K * 4 + rows_small * 48),
         "gather_huge_kernel": ("ids 16 MiB + 4 Mi rows x 48 B = 192 MiB of rows (whole 64-B lines: 470 MB, whole 128-B lines: 738 MB)", K * 4 + K * 48)}
out = {}
for k, d in per.items():
    vals = [d[i] for i in sorted(d)][-2:]
    raw = sum(vals) / len(vals) * 1024.0
    name = k.replace("void ", "")
    what, b = known.get(name, ("?", 0))
    out[name] = {"what": what, "known_min_bytes": b, "FETCH_SIZE_bytes_raw": raw, "raw_over_known": raw / b if b else None,
                 "x2_over_known": 2 * raw / b if b else None, "avg_us": dur.get(k)}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "..", "fetch_probe.json"), "w"), indent=1)
PY
rm -rf "$OUT/fetch" "$OUT/trace"
