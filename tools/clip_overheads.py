"""What the clip fit pays on top of its plain iterations: the same 60-frame clip fitted with and without the reference's
snapshots (every 10th iteration) and per-frame trajectory work, alternating, on one box.  (analysis tool)
    gpurun -- python tools/clip_overheads.py [frames] [rounds]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gflow_amd import synthetic as S, fit_video as FV

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0, device=dev), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000, traj_num=100, traj_offset=2), seed=0, snapshot_interval=10)
torch.cuda.synchronize()
cases = {"snap10_traj100": (10, 100), "snap0_traj100": (0, 100), "snap10_traj0": (10, 0), "snap0_traj0": (0, 0)}
res = {k: [] for k in cases}
its = {}
for r in range(rounds):
    for name, (snap, traj) in cases.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = FV.fit_clip(frames, dev, dict(num_points=60000, traj_num=traj, traj_offset=2), seed=0, snapshot_interval=snap)
        torch.cuda.synchronize(); res[name].append(time.perf_counter() - t0)
        its[name] = m["iterations"]
out = {k: {"wall_s": v, "best_s": min(v), "frames_per_s": n_frames / min(v), "us_per_iteration": min(v) / its[k] * 1e6} for k, v in res.items()}
print(json.dumps(out, indent=1))
