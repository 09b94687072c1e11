import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from gflow_amd import synthetic as S, fit_video as FV
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n, 480, 854, seed=0), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000, traj_num=100, traj_offset=2), seed=0, snapshot_interval=10)
for r in range(3):
    for traj in (0, 100):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = FV.fit_clip(frames, dev, dict(num_points=60000, traj_num=traj, traj_offset=2), seed=0, snapshot_interval=10)
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print(f"traj_num {traj}: {w:.4f} s  ({n / w:.2f} frames/s)", flush=True)
