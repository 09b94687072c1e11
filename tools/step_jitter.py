"""Distribution of the timed step region: ten times 200 steps in one process, each split in chunks of 20 by events.
    gpurun -- python tools/step_jitter.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gflow_amd import synthetic as S, fit_video as FV
from gflow_amd.trainer import SimpleGaussian

dev = torch.device("cuda", 0)
if "--clip-first" in sys.argv:
    frames = S.make_clip(4, bench.H, bench.W, seed=0)
    FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=10)
frame = S.make_frame(bench.H, bench.W, seed=0)
raw = S.init_splats(frame, bench.N_SPLATS, seed=0, grown=True)
tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
tr.load_camera(focal=frame["focal"], pp=frame["pp"])
for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
    tr._attributes[k] = raw[k].to(dev)
st = tr.make_stepper(iterations=500, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
                     move_mask=frame["move_mask"], densify_interval=0, snapshot_interval=0)
st.run(27)
for rep in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for c in range(10):
        st.run(20)
        ev[c + 1].record()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    chunks = [ev[c].elapsed_time(ev[c + 1]) / 20 for c in range(10)]
    print(f"rep {rep}: {dt / 200 * 1e3:.4f} ms/step; per 20-step chunk: " + " ".join(f"{x:.3f}" for x in chunks))
