import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gflow_amd import synthetic as S, fit_video as FV
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(8, 480, 854, seed=0), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=10)
w = []
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=10)
    torch.cuda.synchronize(); w.append(time.perf_counter() - t0)
print("GFL_SNAP_WG", os.environ.get("GFL_SNAP_WG"), "min %.4f median %.4f" % (min(w), float(np.median(w))), np.round(w, 4))
