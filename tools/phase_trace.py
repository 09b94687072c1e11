"""Phase time stamps of the latency-bound launches (preprocess, column scan, scatter, per-splat backward + Adam) on the
bench scene, or with --fit on the joint stage of a clip's second frame (flow and still terms on).  Needs `make TRACE=1`.
    gpurun -- 'make -C gflow_amd/csrc clean; make -C gflow_amd/csrc TRACE=1; python tools/phase_trace.py'
Prints, per kernel, for every phase boundary: mean / p90 / max over the waves of (time since the launch's first stamp)
and of (time since the wave's previous stamp), in microseconds (wall_clock64: 10 ns ticks)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gflow_amd import _lib
from gflow_amd import synthetic as S
from gflow_amd.trainer import SimpleGaussian

dev = torch.device("cuda", 0)
lib = _lib.load()
if "--fit" in sys.argv:
    from gflow_amd import fit_video as FV
    frames = FV.upload_clip(S.make_clip(2, bench.H, bench.W, seed=0), dev)
    FV.fit_clip(frames, dev, dict(num_points=bench.N_SPLATS), seed=0, snapshot_interval=0)
else:
    frame = S.make_frame(bench.H, bench.W, seed=0)
    raw = S.init_splats(frame, bench.N_SPLATS, seed=0, grown=True)
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
        tr._attributes[k] = raw[k].to(dev)
    stepper = tr.make_stepper(iterations=500, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
                              move_mask=frame["move_mask"], densify_interval=0, snapshot_interval=0)
    for _ in range(200):
        stepper()
torch.cuda.synchronize()
NW = 4096
buf = (ctypes.c_longlong * (4 * NW * 8))()
fn = lib.gfl_debug_read_phase_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 4 * NW * 8) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(4, NW, 8)
NAMES = {0: ("preprocess", ["start", "hist zeroed", "row arrived", "project+cov3d", "ewa+hist+stores", "wide walk",
                            "barrier", "hist row out"]),
         1: ("preprocess + binning (reserved tile regions; the exact path's column scan: start, column sums, barrier, bases out)",
             ["start", "hist zeroed", "preprocess", "regions reserved", "barrier", "narrow scatter", "wide walk"]),
         2: ("scatter", ["start", "tile scan", "cursor built", "barrier", "rec arrived", "narrow scatter", "wide walk"]),
         3: ("pre_bwd_adam", ["start", "rows arrived", "gather", "wide gather", "camera", "chain rule", "adam out", "reduced"])}
for k, (name, labels) in NAMES.items():
    t = a[k]
    live = t[:, 0] > 0
    if not live.any():
        continue
    t = t[live]
    t = t[t[:, 0] > t[:, 0].max() - 20000]        # the last launch only (200 us)
    base = t[:, 0].min()
    if k == 3:
        # the scheduling workgroups behind the per-splat ones stamp only their start and their end
        sched = (t[:, 1] == 0) & (t[:, 7] > 0)
        names3 = ["backward queues", "forward queues", "regions + sort order"]
        for b in range(int(sched.sum()) // 4):
            w = t[sched][4 * b:4 * b + 4]
            print(f"  scheduling workgroup {b} ({names3[b] if b < 3 else '?'}): start {(w[:, 0].min() - base) / 100.0:.2f} us, "
                  f"end {(w[:, 7].max() - base) / 100.0:.2f} us")
        print(f"  launch span with them {(t[:, :8].max() - base) / 100.0:.2f} us")
        t = t[~sched]
    print(f"== {name}: {len(t)} waves, launch span {(t[:, :len(labels)].max() - base) / 100.0:.2f} us")
    prev = t[:, 0]
    for j, lab in enumerate(labels):
        col = t[:, j]
        ok = col > 0
        if not ok.any():
            continue
        since = (col[ok] - base) / 100.0
        step = (col[ok] - prev[ok]) / 100.0
        print(f"  {lab:18s} at mean {since.mean():6.2f}  p90 {np.percentile(since, 90):6.2f}  max {since.max():6.2f}"
              f"   | phase mean {step.mean():5.2f}  p90 {np.percentile(step, 90):5.2f}  max {step.max():5.2f}   ({ok.sum()} waves)")
        prev = np.where(ok, col, prev)
