"""Per-tile timeline of the backward (or, with --fwd, forward) blend; --fit: on a real first-frame fit
instead of the bench scene (needs a library built with `make TRACE=1`).
    gpurun -- python tools/bwd_trace.py [--fwd] [--fit]
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
REAL = "--fit" in sys.argv      # a real first-frame fit (image-driven init, densification) instead of the bench scene
if REAL:
    from gflow_amd.fit_video import DEFAULTS as c
    frame = S.make_clip(1, bench.H, bench.W, seed=0)[0]
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    tr.init_gaussians_from_image(frame["image"], frame["depth"], num_points=bench.N_SPLATS)
    tr.train(iterations=c["iterations_first"], lr=c["lr"], lr_camera=c["lr_camera"], lambda_var=c["lambda_var"],
             lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"], densify_interval=c["densify_interval"],
             densify_times=c["densify_times"], move_mask=frame["move_mask"])
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
    for _ in range(200):
        stepper()
torch.cuda.synchronize()
T = tr.engine.T
NT = 16384
buf = (ctypes.c_longlong * (NT * 8))()
FWD = "--fwd" in sys.argv     # the forward blend instead of the backward
fn = lib.gfl_debug_read_fwd_trace if FWD else lib.gfl_debug_read_bwd_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, NT) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(NT, 8)
ROWID = np.nonzero(a[:, 0] > a[:, 1].max() - 30000)[0]     # tile + 2048 * (1 + part) for the near segments of heavy tiles
a = a[a[:, 0] > a[:, 1].max() - 30000]     # last launch only; rows 8192+ are the near halves of split tiles
np.save(os.path.join(ROOT, "gpurun_out", "bwd_trace.npy"), a)
t0, t1 = a[:, 0], a[:, 1]
base = t0.min()
dur = (t1 - t0) / 100.0          # wall_clock64 ticks at 100 MHz -> us
total, depth_n = a[:, 2] >> 32, a[:, 2] & 0xffffffff
units = a[:, 4:8] & 0xffffffff
probes = units
print("kernel span us", (t1.max() - base) / 100.0)
print("tile dur us: mean %.1f max %.1f" % (dur.mean(), dur.max()))
print("start offsets us: max", ((t0 - base) / 100.0).max())
order = np.argsort(-dur)[:15]
for i in order:
    print("tile %4d total %4d depth_n %4d units %s probes %s start %.1f dur %.1f" %
          (i, total[i], depth_n[i], units[i], probes[i], (t0[i] - base) / 100.0, dur[i]))
print("sum units", units.sum(), "sum probes", probes.sum(), "max wave units", units.max())
print("corr(dur, max wave units)", np.corrcoef(dur, units.max(1))[0, 1], "corr(dur, total)", np.corrcoef(dur, total)[0, 1])
# ---- per-CU view
import collections
hw = a[:, 3] & 0xffffffff
xcc = a[:, 3] >> 32
cu = (xcc << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15)
groups = collections.defaultdict(list)
for i, c_ in enumerate(cu):
    groups[c_].append(i)
rows = []
for c_, ts in groups.items():
    rows.append((max((t1[ts] - base) / 100.0), int(units[ts].sum()), len(ts), sorted(units[ts].sum(1).tolist(), reverse=True)[:4],
                 int(units[ts].max())))
rows.sort(reverse=True)
for r in rows[:6]:
    print("CU end %.1f  units %d  items %d  top items %s  max wave %d" % r)
for r in rows[-2:]:
    print("CU end %.1f  units %d  items %d  top items %s  max wave %d" % r)
if "--queues" in sys.argv and not FWD:
    # planned (scheduler's queues) against observed (hardware CU ids): units per queue and which CUs ran a queue's tiles
    qs = tr.engine.schedule()
    tile_units = np.zeros(T, dtype=np.int64)
    tile_cu = [set() for _ in range(T)]
    for i in range(len(ROWID)):
        t = int(ROWID[i]) % 2048
        tile_units[t] += int(units[i].sum())
        tile_cu[t].add(int(cu[i]))
    np.save(os.path.join(ROOT, "gpurun_out", "tile_units.npy"), tile_units)
    np.save(os.path.join(ROOT, "gpurun_out", "queue_tiles.npy"), np.array([np.pad(q.numpy(), (0, 64 - len(q)), constant_values=-1) for q in qs]))
    qsum = np.array([int(tile_units[q.numpy()].sum()) for q in qs])
    print("queues %d: units per queue mean %.0f max %d min %d" % (len(qs), qsum.mean(), qsum.max(), qsum.min()))
    ncu = [len(set().union(*[tile_cu[int(t)] for t in q.numpy()])) if len(q) else 0 for q in qs]
    print("distinct CUs that ran a queue's tiles: mean %.2f max %d" % (np.mean(ncu), max(ncu)))
    # how many queues did a CU serve?
    cu_q = collections.defaultdict(set)
    for qi, q in enumerate(qs):
        for t in q.numpy():
            for c_ in tile_cu[int(t)]:
                cu_q[c_].add(qi)
    nq_per_cu = [len(v) for v in cu_q.values()]
    print("queues per CU: mean %.2f max %d ; CUs seen %d" % (np.mean(nq_per_cu), max(nq_per_cu), len(cu_q)))
if "--items" in sys.argv:
    # item timelines of the three slowest CUs and of a median one
    idx_rows = np.nonzero(a[:, 0] > a[:, 1].max() - 30000)[0] if False else None
    order_cu = sorted(groups.items(), key=lambda kv: -max((t1[kv[1]] - base) / 100.0))
    for c_, ts in order_cu[:3] + [order_cu[len(order_cu) // 2]]:
        print("CU %x: end %.1f" % (c_, max((t1[ts] - base) / 100.0)))
        for i in sorted(ts, key=lambda i: t0[i]):
            print("   row %5d  start %5.1f end %5.1f  entries %4d depth_n %4d  units %s" %
                  (ROWID[i], (t0[i] - base) / 100.0, (t1[i] - base) / 100.0, total[i], depth_n[i], units[i].tolist()))
U = np.array([r[1] for r in rows]); E = np.array([r[0] for r in rows])
# what predicts a CU's finishing time?  least squares  end ~ a units + b list entries + c items + d
L = np.array([int(total[ts].sum()) for ts in groups.values()], dtype=np.float64)
I = np.array([len(ts) for ts in groups.values()], dtype=np.float64)
Uu = np.array([int(units[ts].sum()) for ts in groups.values()], dtype=np.float64)
Ee = np.array([max((t1[ts] - base) / 100.0) for ts in groups.values()])
for names, cols in ((("units",), [Uu]), (("units", "items"), [Uu, I]), (("units", "entries"), [Uu, L]),
                    (("units", "entries", "items"), [Uu, L, I])):
    A = np.stack(cols + [np.ones_like(Uu)], axis=1)
    coef, res, *_ = np.linalg.lstsq(A, Ee, rcond=None)
    pred = A @ coef
    print("fit end ~", " + ".join(f"{c:.4f} {n}" for c, n in zip(coef, names)), f"+ {coef[-1]:.1f}",
          f"| residual rms {np.sqrt(np.mean((pred - Ee) ** 2)):.2f} us, corr {np.corrcoef(pred, Ee)[0, 1]:.2f}")
print("CU units mean %.0f max %d min %d ; end mean %.1f min %.1f max %.1f ; corr(end, units) %.2f" %
      (U.mean(), U.max(), U.min(), E.mean(), E.min(), E.max(), np.corrcoef(E, U)[0, 1]))
if not FWD:
    # per-SIMD view: which SIMD ran wave k of each item, and does the busiest SIMD predict the CU's finishing time?
    simd = (a[:, 4:8] >> 60) & 3
    print("wave k -> SIMD histogram (rows: wave 0..3):")
    for k in range(4):
        print("  wave", k, np.bincount(simd[:, k], minlength=4))
    smax, ssum = [], []
    for ts in groups.values():
        load = np.zeros(4)
        for i in ts:
            for k in range(4):
                load[simd[i, k]] += units[i, k]
        smax.append(load.max()); ssum.append(load.sum())
    smax, ssum = np.array(smax), np.array(ssum)
    print("busiest SIMD units: mean %.0f max %.0f ; (CU units / 4: mean %.0f)" % (smax.mean(), smax.max(), ssum.mean() / 4))
    for names, cols in ((("max SIMD units",), [smax]), (("max SIMD units", "CU units"), [smax, ssum]),
                        (("max SIMD units", "CU units", "entries"), [smax, ssum, L])):
        A = np.stack(cols + [np.ones_like(smax)], axis=1)
        coef, *_ = np.linalg.lstsq(A, Ee, rcond=None)
        pred = A @ coef
        print("fit end ~", " + ".join(f"{c:.4f} {n}" for c, n in zip(coef, names)), f"+ {coef[-1]:.1f}",
              f"| residual rms {np.sqrt(np.mean((pred - Ee) ** 2)):.2f} us, corr {np.corrcoef(pred, Ee)[0, 1]:.2f}")
    lanes = ((a[:, 4:8] >> 32) & 0x0fffffff).sum()
    print("valid lanes per unit: %.1f of 64" % (lanes / max(units.sum(), 1)))
if FWD:
    done_px = (a[:, 4:8] >> 32).sum(1)
    heavy = total > np.percentile(total, 95)
    print("terminated pixels per tile (of 256): all tiles mean %.1f ; 5%% longest lists mean %.1f max %d" %
          (done_px.mean(), done_px[heavy].mean(), done_px.max()))

if FWD and hasattr(lib, "gfl_debug_read_fwd_trace2"):
    # the sixteen quarter waves of the long first tiles: where does a wave's time go?
    n2 = 4096 * 16 * 4
    buf2 = (ctypes.c_longlong * n2)()
    f2 = lib.gfl_debug_read_fwd_trace2
    f2.restype = ctypes.c_int
    f2.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert f2(buf2, n2) == 0
    b = np.frombuffer(buf2, dtype=np.int64).reshape(4096, 16, 4)
    live = np.nonzero(b[:, :, 1].sum(1) > 0)[0]
    print("long first tiles walked as blocks:", len(live))
    tr_rng = tr.engine.tile_range.cpu().numpy()
    for t in sorted(live, key=lambda t: -(b[t, :, 0] + b[t, :, 1]).max())[:8]:
        w = int(np.argmax(b[t, :, 0] + b[t, :, 1]))
        print("  tile %4d list %4d | slowest wave %2d: staging %.1f us, walk %.1f us, steps %d, units %d | mean over 16 waves: staging %.1f walk %.1f steps %.0f"
              % (t, tr_rng[t, 1] - tr_rng[t, 0], w, b[t, w, 0] / 100.0, b[t, w, 1] / 100.0, b[t, w, 2], b[t, w, 3],
                 b[t, :, 0].mean() / 100.0, b[t, :, 1].mean() / 100.0, b[t, :, 2].mean()))
