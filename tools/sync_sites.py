"""Which lines of a clip fit stop the host until the device is idle?  torch's sync debug mode turns every synchronising call
(.item(), .tolist(), .cpu(), bool(tensor), copies from pageable memory ...) into a warning; the call sites inside gflow_amd
are counted per frame stage.   (analysis tool)      gpurun -- python tools/sync_sites.py [frames] [snapshot_interval] [traj]"""
import os, sys, warnings, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
snap = int(sys.argv[2]) if len(sys.argv) > 2 else 10
traj = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0, device=dev), dev)
cfg = dict(num_points=60000, traj_num=traj, traj_offset=2)
FV.fit_clip(frames[:2], dev, cfg, seed=0, snapshot_interval=snap)
torch.cuda.synchronize()
sites = collections.Counter()
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        st = [f for f in traceback.extract_stack() if "gflow_amd" in f.filename]
        key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}({f.name})" for f in reversed(st[-3:]))
        sites[key] += 1
    else:
        orig(message, category, filename, lineno, file, line)


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
FV.fit_clip(frames, dev, cfg, seed=0, snapshot_interval=snap)
torch.cuda.set_sync_debug_mode(0)
torch.cuda.synchronize()
print(f"{n_frames} frames, synchronising call sites (count):")
for k, c in sites.most_common():
    print(f"{c:5d}  {k}")
