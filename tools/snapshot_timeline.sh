#!/bin/bash
# What the host and the device do around a snapshot's device-to-host copy (a 2-frame clip fit under rocprofv3 with the HIP
# API and kernel traces):  gpurun -- 'bash tools/snapshot_timeline.sh'
ROOT=$(pwd); OUT=$ROOT/gpurun_out/snaptl; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$OUT" -o r -- python $ROOT/tools/profile_clip.py 2 10 > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
kern = [r for p in glob.glob(out + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(p))]
api = [r for p in glob.glob(out + "/**/*hip_api_trace.csv", recursive=True) for r in csv.DictReader(open(p))]
print(len(kern), "kernels", len(api), "api calls")
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r["Queue_Id"], r["Kernel_Name"].split("(")[0][-30:])) for r in kern]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A t%s %s" % (r["Thread_Id"][-4:], r["Function"])) for r in api]
ev.sort()
copies = [e for e in ev if "copyBuffer" in e[2] and e[1] - e[0] > 40000]
print(len(copies), "long copy kernels")
for c in copies[30:32]:
    lo, hi = c[0] - 120000, c[1] + 60000
    print("---- copy from %.1f to %.1f us" % ((c[0] - lo) / 1e3, (c[1] - lo) / 1e3))
    for e in ev:
        if e[1] >= lo and e[0] <= hi and (e[1] - e[0] > 3000 or e[2].startswith("K") or "Launch" in e[2] or "Memcpy" in e[2] or "Event" in e[2] or "Wait" in e[2]):
            print("%9.1f %9.1f  %8.1f  %s" % ((e[0] - lo) / 1e3, (e[1] - lo) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
PY
