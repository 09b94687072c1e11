#!/bin/bash
# Profile the default bench workload on the GPU box:
# per pinned window of bench.py (first_frame / camera / joint):
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
# the DRIVER's flags: what BENCH_rNN.json is timed with (bench.py pins the timed windows of the fits whatever the flags are).
# One set of four passes PER PINNED WINDOW (--only-window: nothing else runs, so the last 40 launches of every kernel in a
# trace are two passes over that window): first_frame = the first-frame fit on the grown scene (backward blend <10>),
# camera / joint = frame 30 of the metric's clip after a real fit of the frames before it (<6> + forward <3> + camera Adam;
# <7>): 98 % of the metric's iterations are of the last two kinds.
cd /tmp
for WIN in first_frame camera joint; do
  BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-stage-pass --repeats 0 --only-window $WIN"
  mkdir -p "$OUT/$WIN"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$WIN/trace" -o r -- $BENCH > "$OUT/$WIN/bench_trace.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$WIN/fetch" -o r -- $BENCH > "$OUT/$WIN/bench_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$WIN/write" -o r -- $BENCH > "$OUT/$WIN/bench_write.log" 2>&1
  # issue counters of the kernels (own pass, counters only)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$WIN/sq" -o r -- $BENCH > "$OUT/$WIN/bench_sq.log" 2>&1
done
# kernel durations of an actual clip fit (image-driven start, densification, camera-only stages, snapshots)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/clip" -o r -- python $ROOT/tools/profile_clip.py 3 10 > "$OUT/clip.log" 2>&1
cd "$ROOT"
# (lane efficiency of a (splat, 8x8 block) unit: bench.py's own `work` block of each window, tools/unit_stats.py -- read from
#  the trace pass's JSON line by the summariser)
python tools/summarise_profile.py "$OUT" > "$OUT/../${TAG}_summary.json"
for WIN in first_frame camera joint; do
  cp $(find "$OUT/$WIN/trace" -name "*kernel_stats.csv" | head -1) "$OUT/../${TAG}_${WIN}_kernel_stats.csv"
done
cp $(find "$OUT/clip" -name "*kernel_stats.csv" | head -1) "$OUT/../${TAG}_clip_fit_kernel_stats.csv"
python - "$OUT/../${TAG}_summary.json" > "$OUT/../${TAG}_pmc_current.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps({"tag": d["tag"], "source": "profiles/%s_summary.json (tools/profile_round.sh: the driver's flags, the launches of each pinned window)" % d["tag"],
                  "windows": {w: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "valu": v["valu"], "instantiations": v["instantiations"]}
                              for w, v in d["windows"].items()}}, indent=1))
PY
head -c 3000 "$OUT/../${TAG}_summary.json"
