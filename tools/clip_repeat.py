"""The bench's clip fit (8 frames of 480x854, 60 000 splats, snapshots every 10th iteration) several times in one process:
min / median wall time (a single run varies by +-4 % on a shared host).   python tools/clip_repeat.py [runs] [frames] [async|sync|-] [snapshot interval]
(async / sync: the snapshots on a side stream from a staged copy, or in the snapshot iteration's own forward)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gflow_amd import synthetic as S, fit_video as FV

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kw = {} if len(sys.argv) <= 3 or sys.argv[3] == "-" else {"async_snapshots": sys.argv[3] == "async"}
SNAP = int(sys.argv[4]) if len(sys.argv) > 4 else 10          # snapshot interval (0: none)
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=SNAP, **kw)
torch.cuda.synchronize()
walls = []
for r in range(runs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=SNAP, **kw)
    torch.cuda.synchronize()
    walls.append(time.perf_counter() - t0)
w = np.array(walls)
print(f"{n_frames} frames, {m['iterations']} iterations: min {w.min():.4f} s ({n_frames / w.min():.2f} frames/s, "
      f"{m['iterations'] / w.min():.0f} it/s)  median {np.median(w):.4f} s  void {m.get('void_iterations')}  all {np.round(w, 4)}")
