"""Where does a clip fit spend its wall time?  (analysis tool)  python tools/profile_clip.py [frames] [snapshot_interval]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV
from gflow_amd import trainer as TR

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
snap = int(sys.argv[2]) if len(sys.argv) > 2 else 10
KW = {"async_snapshots": False} if os.environ.get("PROFILE_SYNC_SNAPSHOTS") == "1" else {}     # (in the snapshot iteration's own forward)
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=snap, **KW)
torch.cuda.synchronize()

log = []
orig_train = TR.SimpleGaussian.train
orig_dens = TR.SimpleGaussian.densify_by_pixels
orig_make = TR.SimpleGaussian.make_stepper

def timed(name, fn):
    def w(self, *a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(self, *a, **k)
        torch.cuda.synchronize(); log.append((name, time.perf_counter() - t0, k.get("iterations"), k.get("camera_only", False)))
        return r
    return w

TR.SimpleGaussian.train = timed("train", orig_train)
TR.SimpleGaussian.densify_by_pixels = timed("densify", orig_dens)
TR.SimpleGaussian.make_stepper = timed("make_stepper", orig_make)
torch.cuda.synchronize(); t0 = time.perf_counter()
m = FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=snap, **KW)
torch.cuda.synchronize(); total = time.perf_counter() - t0
print(f"total {total:.3f} s for {m['iterations']} iterations = {m['iterations']/total:.0f} it/s, {n_frames/total:.2f} frames/s")
agg = {}
for name, dt, it, cam in log:
    key = name if name != "train" else f"train(it={it},camera_only={cam})"
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += dt
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:45s} calls {c:3d}  total {t*1e3:8.1f} ms  per call {t/c*1e3:8.2f} ms")
