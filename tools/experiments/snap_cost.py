import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gflow_amd import synthetic as S, fit_video as FV
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(8, 480, 854, seed=0), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=10)
for si in (10, 0, 10, 0, 10, 0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=si)
    torch.cuda.synchronize(); print("snapshot_interval", si, "wall %.4f s" % (time.perf_counter() - t0))
