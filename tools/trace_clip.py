"""A plain clip fit for a rocprofv3 kernel trace (no wrappers, no extra synchronisation): warm-up fit of two frames, a pause of
0.3 s (the marker gap_analysis.py looks for), then the fit.   (analysis tool)
    rocprofv3 --kernel-trace --output-format csv -d out -o r -- python tools/trace_clip.py [frames] [snapshot_interval] [traj_num]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
snap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
traj = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0, device=dev), dev)
cfg = dict(num_points=60000, traj_num=traj, traj_offset=2)
FV.fit_clip(frames[:2], dev, cfg, seed=0, snapshot_interval=snap)
torch.cuda.synchronize()
time.sleep(0.3)
t0 = time.perf_counter()
m = FV.fit_clip(frames, dev, cfg, seed=0, snapshot_interval=snap)
torch.cuda.synchronize()
w = time.perf_counter() - t0
print(f"fit of {n_frames} frames: {w:.3f} s, {m['iterations']} iterations, {w / m['iterations'] * 1e6:.1f} us per iteration")
