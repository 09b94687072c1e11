"""Per-tile timeline of the tile sort (needs a library built with `make TRACE=1`).
    gpurun -- python tools/sort_trace.py
"""
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
if "--fit" in sys.argv:       # a real first-frame fit (image-driven init, two densifications) instead of the bench scene
    from gflow_amd.fit_video import DEFAULTS as c
    frame = S.make_clip(1, bench.H, bench.W, seed=0)[0]
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    tr.init_gaussians_from_image(frame["image"], frame["depth"], num_points=bench.N_SPLATS)
    tr.train(iterations=c["iterations_first"], lr=c["lr"], lr_camera=c["lr_camera"], lambda_var=c["lambda_var"],
             lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"], densify_interval=c["densify_interval"],
             densify_times=c["densify_times"], move_mask=frame["move_mask"], snapshot_interval=0)
    for _ in range(5):
        tr.engine.iteration()
else:
    frame = S.make_frame(bench.H, bench.W, seed=0)
    raw = S.init_splats(frame, bench.N_SPLATS, seed=0, grown=True)
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
        tr._attributes[k] = raw[k].to(dev)
    stepper = tr.make_stepper(iterations=500, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
                              move_mask=frame["move_mask"], densify_interval=0, snapshot_interval=0)
    for _ in range(100):
        stepper()
torch.cuda.synchronize()
T = tr.engine.T
buf = (ctypes.c_longlong * (T * 4))()
fn = lib.gfl_debug_read_sort_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, T) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(T, 4).astype(np.float64) / 100.0
n = (tr.engine.tile_range[:, 1] - tr.engine.tile_range[:, 0]).cpu().numpy()
base = a[:, 0].min()
print("span us", a[:, 3].max() - base, "last start", a[:, 0].max() - base)
for name, lo, hi in (("load", 0, 1), ("network", 1, 2), ("emit", 2, 3), ("total", 0, 3)):
    d = a[:, hi] - a[:, lo]
    print("%-8s mean %.2f  p50 %.2f  p99 %.2f  max %.2f" % (name, d.mean(), np.median(d), np.percentile(d, 99), d.max()))
print("keys", int(n.sum()), "longest lists", np.sort(n)[-8:])
for lo_n, hi_n in ((0, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 4096)):
    m = (n > lo_n) & (n <= hi_n)
    if m.any():
        print("n in (%d,%d]: tiles %d  start mean %.2f max %.2f  network mean %.2f us  total mean %.2f  end mean %.2f max %.2f" %
              (lo_n, hi_n, m.sum(), (a[m, 0] - base).mean(), (a[m, 0] - base).max(), (a[m, 2] - a[m, 1]).mean(),
               (a[m, 3] - a[m, 0]).mean(), (a[m, 3] - base).mean(), (a[m, 3] - base).max()))
