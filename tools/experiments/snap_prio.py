import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gflow_amd import synthetic as S, fit_video as FV
dev = torch.device("cuda", 0)
try:
    print("priority range", torch.cuda.Stream.priority_range())
except Exception as e:
    print("no priority_range:", e)
frames = FV.upload_clip(S.make_clip(8, 480, 854, seed=0), dev)
main_prio = int(os.environ.get("GFL_EXP_MAIN_PRIO", "0"))
ms = torch.cuda.Stream(device=dev, priority=main_prio) if main_prio else torch.cuda.current_stream()
with torch.cuda.stream(ms):
    FV.fit_clip(frames[:2], dev, dict(num_points=60000), seed=0, snapshot_interval=10)
    w = []
    for r in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=10)
        torch.cuda.synchronize(); w.append(time.perf_counter() - t0)
print("side", os.environ.get("GFL_EXP_SIDE_PRIO"), "main", main_prio, "walls", np.round(w, 4), "min %.4f" % min(w))
