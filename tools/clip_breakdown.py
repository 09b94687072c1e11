"""Wall-time breakdown of a clip fit with a device synchronisation around every part (analysis tool; the
synchronisations themselves cost a little: compare the parts, not the total).
    python tools/clip_breakdown.py [frames] [snapshot_interval]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV
from gflow_amd import trainer as TR
from gflow_amd import fused as FU

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
snap = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=snap)
torch.cuda.synchronize()

acc = {}


def add(k, dt, n=1):
    a = acc.setdefault(k, [0, 0.0])
    a[0] += n
    a[1] += dt


def timed(name, fn, key=None):
    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        add(key(*a, **k) if key else name, time.perf_counter() - t0)
        return r
    return w


TR.SimpleGaussian.train = timed("train", TR.SimpleGaussian.train,
                                key=lambda self, **k: f"train total (it={k.get('iterations')}, cam={k.get('camera_only', False)})")
TR.SimpleGaussian.make_stepper = timed("  make_stepper", TR.SimpleGaussian.make_stepper)
TR.SimpleGaussian.densify_by_pixels = timed("    densify_by_pixels", TR.SimpleGaussian.densify_by_pixels)
TR._Stepper.run = timed("  stepper.run", TR._Stepper.run)
FU.FitEngine.snapshot = timed("    snapshot (3 images, device)", FU.FitEngine.snapshot)
orig_it = FU.FitEngine.iteration


def it(self, use_graph=False, count=1, snapshot=False):
    gkey = ("snap", count) if snapshot else count
    will_capture = use_graph and self._launched and not FU.PROFILE["mask"] and (
        self._graph_key != bytes(self.state()) + bytes(self.hp) or gkey not in self._graphs)
    if will_capture:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = orig_it(self, use_graph, count, snapshot)
        torch.cuda.synchronize()
        add(f"    graph capture + first replay (count={count}, snapshot={snapshot})", time.perf_counter() - t0)
        return r
    return orig_it(self, use_graph, count, snapshot)


FU.FitEngine.iteration = it

torch.cuda.synchronize()
t0 = time.perf_counter()
m = FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=snap)
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"total {total:.3f} s for {m['iterations']} iterations = {m['iterations']/total:.0f} it/s, {n_frames/total:.2f} frames/s")
for k, (c, t) in sorted(acc.items(), key=lambda kv: kv[0].strip()):
    print(f"{k:52s} calls {c:4d}  total {t*1e3:8.1f} ms  per call {t/c*1e3:8.3f} ms")
tr = sum(t for k, (c, t) in acc.items() if k.startswith("train total"))
print(f"outside train(): {(total - tr)*1e3:.1f} ms")
