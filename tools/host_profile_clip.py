"""Host-side (Python) profile of a clip fit: where does the interpreter spend its time?  (analysis tool)
    gpurun -- python tools/host_profile_clip.py [frames] [snapshot_interval]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
snap = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0), dev)
FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=snap)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=snap)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
