"""Stage times (library's HIP events) of the camera-only iterations of a clip's second frame.
    gpurun -- python tools/camera_stage_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gflow_amd import synthetic as S, fit_video as FV
from gflow_amd import _lib
from gflow_amd.fused import set_profile
from gflow_amd.trainer import SimpleGaussian

dev = torch.device("cuda", 0)
lib = _lib.load()
c = FV.DEFAULTS
frames = FV.upload_clip(S.make_clip(2, bench.H, bench.W, seed=0), dev)
f0, f1 = frames
tr = SimpleGaussian(f0["image"], f0["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
tr.load_camera(focal=f0["focal"], pp=f0["pp"])
tr.init_gaussians_from_image(f0["image"], f0["depth"], num_points=bench.N_SPLATS)
tr.train(iterations=c["iterations_first"], lr=c["lr"], lr_camera=c["lr_camera"], lambda_var=c["lambda_var"],
         lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"], densify_interval=c["densify_interval"],
         densify_times=c["densify_times"], move_mask=f0["move_mask"], snapshot_interval=0)
tr.set_gt_image(f1["image"]); tr.set_gt_depth(f1["depth"]); tr.set_gt_flow(f0["flow"])
st = tr.make_stepper(iterations=150, lr_camera=c["lr_camera_after"], lambda_var=0.0, lambda_still=0.0,
                     lambda_flow=c.get("lambda_flow", 0.01), lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"],
                     densify_interval=0, camera_only=True, move_mask=f1["move_mask"], snapshot_interval=0)
st.run(20)
torch.cuda.synchronize()
set_profile((1 << len(bench.STAGES)) - 1)
for _ in range(100):
    st()
torch.cuda.synchronize()
set_profile(0)
k = bench.profile_read(lib)
print("camera-only iteration, N =", tr.engine.N, {a: round(b * 1e3, 1) for a, b in k.items()}, "sum", round(sum(k.values()) * 1e3, 1))
