"""Tile list lengths during a real first-frame fit (init_gaussians_from_image, densification on).
    gpurun -- python tools/tile_stats_fit.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gflow_amd import synthetic as S
from gflow_amd.trainer import SimpleGaussian
from gflow_amd.fit_video import DEFAULTS as c

dev = torch.device("cuda", 0)
f0 = S.make_clip(1, 480, 854, seed=0)[0]
tr = SimpleGaussian(f0["image"], f0["depth"], num_points=60000, device=dev, seed=0)
tr.load_camera(focal=f0["focal"], pp=f0["pp"])
tr.init_gaussians_from_image(f0["image"], f0["depth"], num_points=60000)
stepper = tr.make_stepper(iterations=500, lr=c["lr"], lr_camera=c["lr_camera"], lambda_var=c["lambda_var"],
                          lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"], densify_interval=0,
                          move_mask=f0["move_mask"])
done = 0
for stage, upto in (("init", 1), ("after 100", 100), ("after 500", 500)):
    while done < upto:
        stepper()
        done += 1
    eng = tr.engine
    rng = eng.tile_range.cpu().numpy()
    n = rng[:, 1] - rng[:, 0]
    print(stage, "N", eng.N, "K", n.sum(), "mean", n.mean(), "p50/p90/p99/max", np.percentile(n, [50, 90, 99, 100]),
          "tiles>256", (n > 256).sum(), ">512", (n > 512).sum(), ">1024", (n > 1024).sum(), ">2048", (n > 2048).sum())
    rec = eng.rec[:eng.N].cpu().numpy()
    rad = rec[:, 11].view(np.int32)
    print("   radius px: mean %.1f p50 %.0f p99 %.0f max %d ; visible %d" % (rad[rad > 0].mean(), np.percentile(rad[rad > 0], 50), np.percentile(rad[rad > 0], 99), rad.max(), (rad > 0).sum()))
