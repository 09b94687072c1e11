"""Fold the three rocprofv3 passes tools/profile_round.sh wrote into one JSON summary:
per-kernel average duration (kernel-trace --stats) and HBM bytes per launch (FETCH_SIZE and
WRITE_SIZE passes; FETCH_SIZE doubled, the gfx950 correction of MI355X_MICROARCH.md 'HBM').

    python tools/summarise_profile.py gpurun_out/<tag>  > gpurun_out/<tag>_summary.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STAGE_OF = {                       # kernel -> bench.py stage name
    "fused_preprocess_fwd_kernel": "preprocess", "bin_colscan_kernel": "colscan",
    "fused_scatter_kernel": "scatter", "bin_tile_sort_kernel": "tile_sort",
    "fused_blend_fwd_kernel": "blend_fwd", "ssim_stats_kernel": "loss", "loss_grad_kernel": "loss",
    "fused_blend_bwd_kernel": "blend_bwd", "fused_preprocess_bwd_adam_kernel": "pre_bwd_adam",
    "fused_camera_adam_kernel": "camera",
}


def short(name):
    name = name.split("(")[0].split("<")[0]
    return name.replace("gfl::", "").replace("void ", "").strip()


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def kernel_stats(root):
    path = find(root, "*kernel_stats.csv")
    out = {}
    if path:
        for r in csv.DictReader(open(path)):
            # template instances of one kernel (ssim_stats_kernel<0> / <1> / <2>) are folded into one entry
            k, calls, tot = short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"])
            e = out.setdefault(k, {"calls": 0, "avg_us": 0.0, "pct": 0.0})
            e["avg_us"] = (e["avg_us"] * e["calls"] + tot / 1e3) / (e["calls"] + calls)
            e["calls"] += calls
            e["pct"] += float(r["Percentage"])
    return out


def counter_avg(root, counter):
    path = find(root, "*counter_collection.csv")
    tot, cnt = defaultdict(float), defaultdict(int)
    if path:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    root = sys.argv[1]
    stats = kernel_stats(os.path.join(root, "trace"))
    fetch = counter_avg(os.path.join(root, "fetch"), "FETCH_SIZE")
    write = counter_avg(os.path.join(root, "write"), "WRITE_SIZE")
    # FETCH_SIZE / WRITE_SIZE are reported in KB
    kernels, stage_bytes = {}, defaultdict(float)
    for k, s in stats.items():
        if k not in STAGE_OF:
            continue
        rd = 2.0 * 1024.0 * fetch.get(k, 0.0)
        wr = 1024.0 * write.get(k, 0.0)
        kernels[k] = dict(s, hbm_read_bytes=rd, hbm_write_bytes=wr, hbm_bytes=rd + wr,
                          hbm_GBps=(rd + wr) / (s["avg_us"] * 1e-6) / 1e9)
        stage_bytes[STAGE_OF[k]] += rd + wr
    sq = {}
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
              "SQ_ACTIVE_INST_LDS", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
        for k, v in counter_avg(os.path.join(root, "sq"), c).items():
            if k in STAGE_OF:
                sq.setdefault(k, {})[c] = v
    clip = {k: v for k, v in kernel_stats(os.path.join(root, "clip")).items() if k in STAGE_OF or "blend" in k}
    print(json.dumps({
        "tag": os.path.basename(os.path.normpath(root)),
        "command": "python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-stage-pass --no-clip",
        "method": "rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE (three separate runs); "
                  "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), both counters in KB",
        "kernels": kernels,
        "hbm_bytes_per_launch": dict(stage_bytes),
        "sq_counters_per_launch": sq,
        "sq_note": "SQ_* summed over the chip per launch; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY count "
                   "quad-cycles, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs (MI355X_MICROARCH.md)",
        "clip_fit_kernels": clip,
        "clip_fit_note": "kernel durations inside an actual 3-frame clip fit (python tools/profile_clip.py 3 10): heavy "
                         "tiles after densification make the blend kernels and the tile sort slower than on the bench scene",
    }, indent=1))


if __name__ == "__main__":
    main()
