import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from gflow_amd import synthetic as S, fit_video as FV, fused
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(8, 480, 854, seed=0, device=dev), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=10)
torch.cuda.synchronize()
orig_begin = torch.cuda.CUDAGraph.capture_begin
orig_end = torch.cuda.CUDAGraph.capture_end
stat = {"n": 0, "t": 0.0, "t0": 0.0}
def b(self, *a, **k):
    stat["t0"] = time.perf_counter(); return orig_begin(self, *a, **k)
def e(self, *a, **k):
    r = orig_end(self, *a, **k); stat["t"] += time.perf_counter() - stat["t0"]; stat["n"] += 1; return r
torch.cuda.CUDAGraph.capture_begin = b
torch.cuda.CUDAGraph.capture_end = e
# also time the first replay (instantiate/upload happens there?)
orig_replay = torch.cuda.CUDAGraph.replay
rs = {"n": 0, "t": 0.0}
def r(self):
    t0 = time.perf_counter(); out = orig_replay(self); rs["t"] += time.perf_counter() - t0; rs["n"] += 1; return out
torch.cuda.CUDAGraph.replay = r
t0 = time.perf_counter()
m = FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=10)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("wall %.3f s; captures %d, %.1f ms total (%.2f ms each); replays %d, %.1f ms host total" % (wall, stat["n"], stat["t"] * 1e3, stat["t"] * 1e3 / max(stat["n"], 1), rs["n"], rs["t"] * 1e3))
