"""Fold the rocprofv3 passes tools/profile_round.sh wrote into one JSON summary:
per-kernel average duration (kernel-trace) and HBM bytes per launch (FETCH_SIZE and
WRITE_SIZE passes; FETCH_SIZE doubled, the gfx950 correction of MI355X_MICROARCH.md 'HBM'), the SQ issue counters.

Every figure is an average over the launches of ONE of bench.py's three PINNED WINDOWS (first_frame / camera / joint: one
directory of passes each, bench.py --only-window) -- the last WINDOW_LAUNCHES dispatches of each kernel: with the driver's
flags (--steps 20 --warmup 5 --no-stage-pass --repeats 0) bench.py runs up to the window, 27 warm-up steps, and then the
window twice (the timed pass and the pass that reads the pair counts) --, so the counters belong to the workload the bench
line is timed on.  The whole-run --stats table is kept beside it.

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
    # reserved tile regions: ONE launch in place of the three above in every iteration that follows a full iteration (19 of
    # the window's 20: the first after restore_state takes the exact path, so the "last 40 launches" of the three exact-path
    # kernels reach back in front of the window)
    "fused_preprocess_bin_kernel": "preprocess_bin",
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


WINDOW_LAUNCHES = 40


def window_durations(root, last=WINDOW_LAUNCHES):
    """average duration (us) over the last ``last`` dispatches of every kernel, from the per-dispatch kernel trace"""
    path = find(root, "*kernel_trace.csv")
    per = defaultdict(list)
    if path:
        for r in csv.DictReader(open(path)):
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out = {}
    for k, v in per.items():
        v.sort()
        w = v[-last:]
        out[k] = {"calls": len(w), "avg_us": sum(e - b for b, e in w) / len(w) / 1e3}
    return out


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


def counter_avg(root, counter, last=WINDOW_LAUNCHES):
    """average of ``counter`` over the last ``last`` dispatches of every kernel (a dispatch's value may come in several
    rows -- one per XCD / instance --: they are summed per dispatch first)"""
    path = find(root, "*counter_collection.csv")
    per = defaultdict(lambda: defaultdict(float))
    if path:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                per[short(r["Kernel_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {}
    for k, d in per.items():
        vals = [d[i] for i in sorted(d)][-last:]
        out[k] = sum(vals) / len(vals)
    return out


def full_names(root, last=WINDOW_LAUNCHES):
    """the template instantiations among the last ``last`` dispatches of every (folded) kernel name"""
    path = find(root, "*kernel_trace.csv")
    per = defaultdict(list)
    if path:
        for r in csv.DictReader(open(path)):
            nm = r["Kernel_Name"].split("(")[0].replace("gfl::", "").replace("void ", "").strip()
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), nm))
    return {k: sorted({n for _, n in sorted(v)[-last:]}) for k, v in per.items()}


def bench_line(root):
    """the JSON line bench.py printed under the trace pass (its `work` blocks: units and lane efficiency per window)"""
    path = os.path.join(root, "bench_trace.log")
    if not os.path.exists(path):
        return None
    for line in open(path, errors="replace"):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


WORK_KEY = {"first_frame": lambda d: d["config"]["step_window"].get("work"), "camera": lambda d: (d.get("step_window_camera") or {}).get("work"),
            "joint": lambda d: (d.get("step_window_clip") or {}).get("work")}


def one_window(root, name):
    whole_run = kernel_stats(os.path.join(root, "trace"))
    stats = window_durations(os.path.join(root, "trace"))
    for k, v in stats.items():
        v["pct"] = whole_run.get(k, {}).get("pct", 0.0)
    inst = full_names(os.path.join(root, "trace"))
    fetch = counter_avg(os.path.join(root, "fetch"), "FETCH_SIZE")
    write = counter_avg(os.path.join(root, "write"), "WRITE_SIZE")
    # FETCH_SIZE / WRITE_SIZE are reported in KB
    kernels, stage_bytes = {}, defaultdict(float)
    for k, s in stats.items():
        if k not in STAGE_OF:
            continue
        rd = 2.0 * 1024.0 * fetch.get(k, 0.0)
        wr = 1024.0 * write.get(k, 0.0)
        kernels[k] = dict(s, instantiations=inst.get(k), hbm_read_bytes=rd, hbm_write_bytes=wr, hbm_bytes=rd + wr,
                          hbm_GBps=(rd + wr) / (s["avg_us"] * 1e-6) / 1e9)
        stage_bytes[STAGE_OF[k]] += rd + wr
    sq = {}
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
              "SQ_ACTIVE_INST_LDS", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
        for k, v in counter_avg(os.path.join(root, "sq"), c).items():
            if k in STAGE_OF:
                sq.setdefault(k, {})[c] = v
    # the issue bound of every stage: how busy the VALUs were.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over the
    # chip's 1024 SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs.
    valu = {}
    for k, c in sq.items():
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU") is not None:
            st = STAGE_OF[k]
            e = valu.setdefault(st, {"insts_per_launch": 0.0, "busy_cycles": 0.0, "avail_cycles": 0.0})
            e["insts_per_launch"] += c.get("SQ_INSTS_VALU", 0.0)
            e["busy_cycles"] += 4.0 * c["SQ_ACTIVE_INST_VALU"]
            e["avail_cycles"] += 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
    for e in valu.values():
        e["busy_frac"] = e.pop("busy_cycles") / e.pop("avail_cycles")
    work = None
    line = bench_line(root)
    if line:
        try:
            work = WORK_KEY[name](line)
        except (KeyError, TypeError):
            work = None
    if work and "lane_efficiency_fwd" in work:
        if "blend_fwd" in valu:
            valu["blend_fwd"]["lane_efficiency"] = work["lane_efficiency_fwd"]
            valu["blend_fwd"]["insts_per_unit"] = valu["blend_fwd"]["insts_per_launch"] / max(work["units_8x8_fwd"], 1)
        if "blend_bwd" in valu:
            valu["blend_bwd"]["lane_efficiency"] = work["lane_efficiency_bwd"]
            valu["blend_bwd"]["insts_per_unit"] = valu["blend_bwd"]["insts_per_launch"] / max(work["units_8x8_bwd"], 1)
    return {"kernels": kernels, "instantiations": {STAGE_OF[k]: v for k, v in inst.items() if k in STAGE_OF},
            "whole_run_stats": {k: v for k, v in whole_run.items() if k in STAGE_OF},
            "hbm_bytes_per_launch": dict(stage_bytes), "valu": valu, "work": work, "sq_counters_per_launch": sq,
            # (the three exact-path binning kernels run once per 20 iterations of a window -- the last 40 launches of each reach
            #  back in front of it -- and the camera launch does not run in the first-frame and joint windows at all: its last
            #  launches there are the camera-only stage's)
            "iteration_us": sum(v["avg_us"] for k, v in kernels.items()
                                if k not in ("fused_preprocess_fwd_kernel", "bin_colscan_kernel", "fused_scatter_kernel")
                                and not (name != "camera" and k == "fused_camera_adam_kernel")),
            "iteration_us_note": "sum of the window averages of the kernels of an iteration on reserved tile regions (19 of the "
                                 "window's 20): preprocess_bin, tile sort, forward, the loss pair, backward, per-splat + Adam"
                                 + (", camera Adam" if name == "camera" else "")}


def main():
    root = sys.argv[1]
    windows = {w: one_window(os.path.join(root, w), w) for w in ("first_frame", "camera", "joint")
               if os.path.isdir(os.path.join(root, w))}
    clip = {k: v for k, v in kernel_stats(os.path.join(root, "clip")).items() if k in STAGE_OF or "blend" in k}
    print(json.dumps({
        "tag": os.path.basename(os.path.normpath(root)),
        "command": "python bench.py --steps 20 --warmup 5 --no-stage-pass --repeats 0 --only-window <first_frame|camera|joint>  "
                   "(the driver's flags; one pinned window per set of passes)",
        "method": "rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc SQ_* (separate runs); "
                  "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), both counters in KB; every figure averaged over "
                  f"the last {WINDOW_LAUNCHES} launches of the kernel = two passes over that pinned window of bench.py",
        "windows": windows,
        "sq_note": "SQ_* summed over the chip per launch; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY count "
                   "quad-cycles, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs (MI355X_MICROARCH.md)",
        "clip_fit_kernels": clip,
        "clip_fit_note": "kernel durations inside an actual 3-frame clip fit (python tools/profile_clip.py 3 10): heavy "
                         "tiles after densification make the blend kernels and the tile sort slower than on the bench scene",
    }, indent=1))


if __name__ == "__main__":
    main()
