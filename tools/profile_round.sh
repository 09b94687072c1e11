#!/bin/bash
# Profile the default bench workload on the GPU box:
#   pass 1  rocprofv3 --kernel-trace --stats      -> per-kernel durations
#   pass 2  rocprofv3 --pmc FETCH_SIZE            -> HBM read bytes per launch   (own pass)
#   pass 3  rocprofv3 --pmc WRITE_SIZE            -> HBM write bytes per launch  (own pass)
#   pass 4  rocprofv3 --pmc SQ_*                  -> VALU instructions / busy cycles per launch (own pass)
#   pass 5  rocprofv3 --kernel-trace --stats      -> per-kernel durations of a real 3-frame clip fit
# then tools/summarise_profile.py folds the three into gpurun_out/<tag>_summary.json.
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh r01b'
# Copy the summary + kernel stats into profiles/ afterwards (gpurun_out/ is scratch).
set -u
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the DRIVER's flags: what BENCH_rNN.json is timed with (bench.py pins the timed window of the fit whatever the flags are)
BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-pass --no-clip"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- $BENCH > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o r -- $BENCH > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o r -- $BENCH > "$OUT/bench_write.log" 2>&1
# pass 4: issue counters of the kernels (own pass, counters only)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o r -- $BENCH > "$OUT/bench_sq.log" 2>&1
# pass 5: kernel durations of an actual clip fit (image-driven start, densification, camera-only stages, snapshots)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/clip" -o r -- python $ROOT/tools/profile_clip.py 3 10 > "$OUT/clip.log" 2>&1
cd "$ROOT"
# lane efficiency of a (splat, 8x8 block) unit on the window's scene (the second bound's third figure)
python tools/lane_efficiency.py --json "$OUT/lane_efficiency.json" > "$OUT/lane_efficiency.log" 2>&1
python tools/summarise_profile.py "$OUT" > "$OUT/../${TAG}_summary.json"
cp "$OUT"/trace/*kernel_stats.csv "$OUT/../${TAG}_kernel_stats.csv" 2>/dev/null || cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/../${TAG}_kernel_stats.csv"
cp $(find "$OUT/clip" -name "*kernel_stats.csv" | head -1) "$OUT/../${TAG}_clip_fit_kernel_stats.csv"
python - "$OUT/../${TAG}_summary.json" > "$OUT/../${TAG}_pmc_current.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps({"tag": d["tag"], "source": "profiles/%s_summary.json (tools/profile_round.sh: the driver's flags, the pinned window's launches)" % d["tag"],
                  "hbm_bytes_per_launch": d["hbm_bytes_per_launch"], "valu": d["valu"]}, indent=1))
PY
head -c 3000 "$OUT/../${TAG}_summary.json"
