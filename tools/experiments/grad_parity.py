import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_fullsize as TF
from test_gpu_fullsize import *
from gflow_amd.fused import COLS
for which in ("bench_scene", "densified_scene"):
    frame, raw = TF._bench_scene() if which == "bench_scene" else TF._densified_scene()
    n = raw["xyz"].shape[0]
    s = dict(W=TF.W, H=TF.H, intr=raw["intr"])
    lam = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0)
    pose0 = torch.tensor([0.002, -0.001, 0.0015, 1.0, 0.01, -0.02, 0.015])
    eng = TF._engine({k: raw[k] for k in TF.NAMES}, s, frame["image"], frame["depth"], pose=pose0, lr=1e-4, lr_camera=1e-4, total_iters=500, **lam)
    eng.iteration()
    rc = {k: raw[k].clone().requires_grad_(True) for k in TF.NAMES}
    pose = pose0.clone().requires_grad_(True); ab = torch.tensor([1.0, 0.0], requires_grad=True)
    torch.set_num_threads(16)
    loss, info = TF.FO.fit_loss(rc, pose, ab, raw["intr"], dict(image=frame["image"], depth=frame["depth"]), 0.0, 1.0, 0.1, 10.0)
    loss.backward()
    g_all = (eng.adam_m[:n] / 0.1).cpu()
    out = {}
    for k, (a, b) in COLS.items():
        ref = rc[k].grad.reshape(n, b - a)
        out[k] = ((g_all[:, a:b] - ref).norm() / ref.norm()).item()
    gp = (eng.pose_m / 0.1).cpu()
    out["pose"] = ((gp - pose.grad).norm() / pose.grad.norm()).item()
    l_rgb, l_depth = eng.loss_terms()
    out["l_rgb"] = abs(l_rgb.item() - info["l_rgb"].item()) / abs(info["l_rgb"].item())
    out["l_depth"] = abs(l_depth.item() - info["l_depth"].item()) / abs(info["l_depth"].item())
    out["K"] = (eng.K, info["K"])
    print(which, {k: (round(v, 7) if isinstance(v, float) else v) for k, v in out.items()})
