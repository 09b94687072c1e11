import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_fused as TF
from gflow_amd.fused import FitEngine
DEV="cuda"
s = TF.random_scene(2500, 168, 120, seed=31, sigma_px=2.5, tilt=False)
raw = TF._raw_from_scene(s); img, dep = TF._targets(s["H"], s["W"], 5)
hyper = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=2e-3, lr_camera=0.0, total_iters=50)
big = TF._engine(raw, s, img, dep, pose=TF.POSE, **hyper)
big.forward(); K = big.K
small = FitEngine(s["W"], s["H"], capacity=big.cap, device=DEV, K_cap=K // 3)
small.set_splats({k: v.to(DEV) for k, v in raw.items()})
small.intr.copy_(s["intr"].to(DEV)); small.pose.copy_(TF.POSE.to(DEV)); small.set_targets(img, dep)
for k, v in hyper.items(): setattr(small.hp, k, v)
small.reset_optimizer()
for rnd in range(3):
    for i in range(3):
        small.iteration(); torch.cuda.synchronize()
        print(rnd, i, small.overflow.tolist(), small.K, small.K_cap, small._reserved_N)
    print("settle", small.settle_overflow())
