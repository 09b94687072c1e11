"""Where does the wall time of a clip fit go?  (host-side profile of gflow_amd.fit_video.fit_clip)
    gpurun -- python tools/profile_fit_video.py [frames]
"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gflow_amd import synthetic as S
from gflow_amd import fit_video as FV
from gflow_amd.trainer import SimpleGaussian

frames_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
frames = S.make_clip(frames_n, 480, 854, seed=0)
print("make_clip s", time.perf_counter() - t0)

orig_train = SimpleGaussian.train
def timed_train(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = orig_train(self, *a, **k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    it = k.get("iterations", a[0] if a else 0)
    print(f"train(iterations={it}, camera_only={k.get('camera_only', False)}): {dt*1000:.1f} ms  ({dt/it*1000:.3f} ms/it)")
    return r
SimpleGaussian.train = timed_train
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0)      # warm-up (library load, graph capture paths)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
m = FV.fit_clip(frames, dev, dict(num_points=60000), seed=0)
torch.cuda.synchronize()
pr.disable()
print("fit_clip s", time.perf_counter() - t0, m)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
