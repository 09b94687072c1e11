"""What does one snapshot cost inside a first-frame fit?  (analysis tool)   gpurun -- python tools/snapshot_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gflow_amd import synthetic as S
from gflow_amd.trainer import SimpleGaussian
from gflow_amd.fit_video import DEFAULTS as c

dev = torch.device("cuda", 0)
frame = S.make_clip(1, bench.H, bench.W, seed=0)[0]

def fit(snap):
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    tr.init_gaussians_from_image(frame["image"], frame["depth"], num_points=bench.N_SPLATS)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = tr.train(iterations=c["iterations_first"], lr=c["lr"], lr_camera=c["lr_camera"], lambda_var=c["lambda_var"],
             lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"], densify_interval=c["densify_interval"],
             densify_times=c["densify_times"], move_mask=frame["move_mask"], snapshot_interval=snap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return tr, dt, out

for snap in (10, 0, 10, 0, 10):
    tr, dt, out = fit(snap)
    print(f"first-frame fit, snapshot_interval={snap}: {dt*1e3:.1f} ms")
eng = tr.engine
for _ in range(3):
    eng.iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    imgs = eng.snapshot()
torch.cuda.synchronize()
print(f"eng.snapshot(): {(time.perf_counter()-t0)/20*1e3:.3f} ms each (device time incl. launch)")
t0 = time.perf_counter()
pin = torch.empty((50, 3, bench.H, bench.W, 3), dtype=torch.uint8, pin_memory=True)
print(f"pinned alloc of {pin.numel()/1e6:.0f} MB: {(time.perf_counter()-t0)*1e3:.1f} ms")
del pin
t0 = time.perf_counter()
pin = torch.empty((50, 3, bench.H, bench.W, 3), dtype=torch.uint8, pin_memory=True)
print(f"second pinned alloc (cached?): {(time.perf_counter()-t0)*1e3:.1f} ms")
cs = torch.cuda.Stream(device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(20):
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        pin[k].copy_(imgs, non_blocking=True)
cs.synchronize()
print(f"copy to pinned: {(time.perf_counter()-t0)/20*1e3:.3f} ms each")
t0 = time.perf_counter()
for _ in range(200):
    eng.iteration()
torch.cuda.synchronize()
print(f"iteration: {(time.perf_counter()-t0)/200*1e3:.3f} ms")
t0 = time.perf_counter()
for i in range(200):
    eng.iteration()
    if i % 10 == 0:
        imgs = eng.snapshot()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            pin[i // 10].copy_(imgs, non_blocking=True)
        imgs.record_stream(cs)
torch.cuda.synchronize()
print(f"iteration with a snapshot every 10: {(time.perf_counter()-t0)/200*1e3:.3f} ms")
from gflow_amd import fit_video as FV
frames = S.make_clip(3, bench.H, bench.W, seed=0)
for snap in (10, 10, 0, 10, 0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = FV.fit_clip(frames[:1], dev, dict(num_points=60000), seed=0, snapshot_interval=snap)
    torch.cuda.synchronize(); print(f"fit_clip 1 frame snap={snap}: {(time.perf_counter()-t0)*1e3:.1f} ms")
for snap in (10, 0, 10, 0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = FV.fit_clip(frames, dev, dict(num_points=60000), seed=0, snapshot_interval=snap)
    torch.cuda.synchronize(); print(f"fit_clip 3 frames snap={snap}: {(time.perf_counter()-t0)*1e3:.1f} ms")
