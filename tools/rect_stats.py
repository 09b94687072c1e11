"""How many tiles does a splat's rectangle cover?  (bench scene and the end of a 3-frame clip fit)
    gpurun -- python tools/rect_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from gflow_amd import synthetic as S, fit_video as FV
from gflow_amd.trainer import SimpleGaussian
from gflow_amd import trainer as TR

dev = torch.device("cuda", 0)


def stats(name, eng):
    rec = eng.rec[:eng.N].cpu().numpy()
    u, v, r = rec[:, 0], rec[:, 1], rec[:, 11].view(np.int32).astype(np.float32)
    gx, gy = eng.gx, eng.gy
    x0 = np.clip(((u - r) / 16).astype(int), 0, gx); x1 = np.clip(((u + r + 15) / 16).astype(int), 0, gx)
    y0 = np.clip(((v - r) / 16).astype(int), 0, gy); y1 = np.clip(((v + r + 15) / 16).astype(int), 0, gy)
    nt = ((x1 - x0) * (y1 - y0))[r > 0]
    print(f"{name}: N {eng.N} visible {len(nt)}  tiles per rectangle: mean {nt.mean():.2f}  "
          + "  ".join(f"<={k}: {100.0 * (nt <= k).mean():.2f}%" for k in (1, 2, 4, 6, 8, 9, 12, 16, 32))
          + f"  max {nt.max()}")


frame = S.make_frame(bench.H, bench.W, seed=0)
raw = S.init_splats(frame, bench.N_SPLATS, seed=0, grown=True)
tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
tr.load_camera(focal=frame["focal"], pp=frame["pp"])
for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
    tr._attributes[k] = raw[k].to(dev)
st = tr.make_stepper(iterations=500, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, move_mask=frame["move_mask"],
                     densify_interval=0, snapshot_interval=0)
st.run(20)
torch.cuda.synchronize()
stats("bench scene", tr.engine)

engines = []
orig = TR.SimpleGaussian.train_steps
def spy(self, *a, **k):
    r = yield from orig(self, *a, **k)
    engines.append(self.engine)
    return r
TR.SimpleGaussian.train_steps = spy
frames = FV.upload_clip(S.make_clip(3, bench.H, bench.W, seed=0), dev)
FV.fit_clip(frames, dev, dict(num_points=bench.N_SPLATS), seed=0, snapshot_interval=0)
torch.cuda.synchronize()
stats("end of a 3-frame clip fit", engines[-1])
