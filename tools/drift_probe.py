"""Fused vs operator path over a long small clip: per-frame PSNR, splat counts, still labels (analysis tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
H, W = 96, 128
dev = torch.device("cuda", 0)
cfg = dict(num_points=1500, iterations_first=60, iterations_after=40, iterations_camera=20, densify_interval=30,
           densify_times=1, densify_interval_after=20, densify_times_after=1, lambda_depth=1e-2)
if "--no-err-densify" in sys.argv:
    cfg.update(densify_interval=0, densify_interval_after=0)
if "--full" in sys.argv:
    cfg = dict(num_points=3000)
for seed in (0, 1):
    frames = FV.upload_clip(S.make_clip(n, H, W, seed=seed, device=dev), dev)
    out = {}
    for fused in (True, False):
        k = {}
        m = FV.fit_clip(frames, dev, cfg, seed=0, fused=fused, keep=k)
        out[fused] = (m, [float(p) for p in k["psnr"]], k["trainer"].still_mask.cpu(), k["trainer"].current_pts_num())
    a, b = out[True], out[False]
    print("seed", seed, "N", a[3], b[3])
    print(" fused   ", " ".join("%.2f" % p for p in a[1]))
    print(" operator", " ".join("%.2f" % p for p in b[1]))
    print(" max |d psnr| %.3f  mean d %.3f" % (max(abs(x - y) for x, y in zip(a[1], b[1])), sum(x - y for x, y in zip(a[1], b[1])) / n))
    if a[3] == b[3]:
        print(" still labels differ on", int((a[2] != b[2]).sum()), "of", a[3], " still share", float(a[2].float().mean()), float(b[2].float().mean()))
    else:
        print(" still share", float(a[2].float().mean()), float(b[2].float().mean()))
