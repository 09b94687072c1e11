"""Distribution of the per-tile list lengths of the bench workload (load-balance analysis).
    gpurun -- python tools/tile_stats.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gflow_amd import synthetic as S
from gflow_amd.trainer import SimpleGaussian

dev = torch.device("cuda", 0)
frame = S.make_frame(bench.H, bench.W, seed=0)
raw = S.init_splats(frame, bench.N_SPLATS, seed=0, grown=True)
tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
tr.load_camera(focal=frame["focal"], pp=frame["pp"])
for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
    tr._attributes[k] = raw[k].to(dev)
stepper = tr.make_stepper(iterations=500, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
                          move_mask=frame["move_mask"], densify_interval=0, snapshot_interval=0)
for _ in range(30):
    stepper()
torch.cuda.synchronize()
eng = tr.engine
rng = eng.tile_range.cpu().numpy()
n = rng[:, 1] - rng[:, 0]
gx = (bench.W + 15) // 16
print("tiles", n.size, "K", n.sum(), "mean", n.mean(), "std", n.std(), "max", n.max(), "min", n.min())
print("percentiles 50/90/99/100:", np.percentile(n, [50, 90, 99, 100]))
srt = np.sort(n)[::-1]
print("top 20:", srt[:20])
# work per CU if tile b goes round-robin to 256 CUs
for name, order in (("plain", np.arange(n.size)), ("sorted", np.argsort(-n))):
    cu = np.zeros(256)
    for b, t in enumerate(order):
        cu[b % 256] += n[t]
    print(name, "CU load mean", cu.mean(), "max", cu.max(), "max/mean", cu.max() / cu.mean())
nc = eng.n_contrib.cpu().numpy().reshape(bench.H, bench.W) if hasattr(eng, "n_contrib") else None
if nc is not None:
    print("n_contrib mean", nc.mean(), "max", nc.max())
np.save(os.path.join(ROOT, "gpurun_out", "tile_counts.npy"), n.reshape(-1, gx))
