import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_fused as TF
from test_gpu_fused import *
s = TF.random_scene(2500, 168, 120, seed=31, sigma_px=2.5, tilt=False)
raw = TF._raw_from_scene(s); img, dep = TF._targets(s["H"], s["W"], 5)
hyper = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=4e-3, lr_camera=0.0, total_iters=100)
a = TF._engine(raw, s, img, dep, pose=TF.POSE, **hyper); b = TF._engine(raw, s, img, dep, pose=TF.POSE, **hyper)
a.iteration(); TF._copy_engine_state(a, b)
print("params equal", torch.equal(a.params[:a.N], b.params[:b.N]))
b.iteration(reserved=False); a.iteration()
torch.cuda.synchronize()
print(a.overflow.tolist(), b.overflow.tolist(), a.K, b.K)
d = (a.rec[:a.N] - b.rec[:b.N]).abs()
print("rec maxdiff per col", d.max(0).values.tolist())
print("nan", torch.isnan(a.rec[:a.N]).sum().item(), torch.isnan(b.rec[:b.N]).sum().item())
bits = (a.rec[:a.N].view(torch.int32) != b.rec[:b.N].view(torch.int32))
print("differing entries per col", bits.sum(0).tolist())
print("render eq", torch.equal(a.render, b.render), (a.render-b.render).abs().max().item())
