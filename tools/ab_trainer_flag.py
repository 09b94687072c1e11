"""A/B of a SimpleGaussian attribute inside clip fits, alternating in ONE process on one box (medians):
   python tools/ab_trainer_flag.py exact_snapshots True False [runs] [frames] [traj_num]
(the attribute is set on every trainer fit_clip creates: SimpleGaussian.__init__ is wrapped)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gflow_amd import synthetic as S, fit_video as FV
from gflow_amd import trainer as T

name, a, b = sys.argv[1], eval(sys.argv[2]), eval(sys.argv[3])
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 4
n_frames = int(sys.argv[5]) if len(sys.argv) > 5 else 8
traj = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0), dev)
cfg = dict(num_points=60000, traj_num=traj, traj_offset=2)
value = [a]
init = T.SimpleGaussian.__init__


def patched(self, *args, **kw):
    init(self, *args, **kw)
    setattr(self, name, value[0])


T.SimpleGaussian.__init__ = patched
FV.fit_clip(frames[:2], dev, cfg, seed=0, snapshot_interval=10)
walls = {repr(a): [], repr(b): []}
for r in range(runs):
    for v in (a, b):
        value[0] = v
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = FV.fit_clip(frames, dev, cfg, seed=0, snapshot_interval=10)
        torch.cuda.synchronize()
        walls[repr(v)].append(time.perf_counter() - t0)
for k, w in walls.items():
    w = np.array(w)
    print(f"{name}={k}: median {np.median(w):.4f} s  min {w.min():.4f} s ({n_frames / np.median(w):.2f} frames/s)  all {np.round(w, 4)}  void {m.get('void_iterations')}")
