"""Host wall time inside the frame-boundary functions of a clip fit (analysis tool).
    gpurun -- python tools/boundary_time.py [frames]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV, trainer as TR, render as RM, fused as FU

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
frames = S.make_clip(n_frames, 480, 854, seed=0)
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=10)
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        e = acc[label or name]; e[0] += 1; e[1] += time.perf_counter() - t0
        return r
    setattr(obj, name, w)
G = TR.SimpleGaussian
for n in ("set_gt_image", "set_gt_depth", "set_gt_flow", "load_camera", "psnr", "make_stepper", "densify_by_pixels", "train",
          "init_gaussians_from_image", "_input_group"):
    if hasattr(G, n): wrap(G, n)
wrap(RM, "render_multiple"); wrap(RM, "render2img")
wrap(FU.FitEngine, "check_overflow"); wrap(FU.FitEngine, "set_targets"); wrap(FU.FitEngine, "snapshot")
wrap(torch.cuda, "empty_cache")
torch.cuda.synchronize(); t0 = time.perf_counter()
FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=10)
torch.cuda.synchronize(); total = time.perf_counter() - t0
print(f"total {total*1e3:.1f} ms for {n_frames} frames")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} calls {c:4d} total {t*1e3:8.2f} ms  per call {t/c*1e3:7.3f} ms")
