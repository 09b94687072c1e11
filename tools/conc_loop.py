import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tests.test_gpu_fitvideo import _clip, SMALL, DEV
from gflow_amd.fit_video import fit_clip, fit_clips_concurrent
clips = [_clip(seed=s) for s in (11, 12, 13)]
for r in range(10):
    alone = [fit_clip(c, DEV, SMALL, seed=i, snapshot_interval=10) for i, c in enumerate(clips)]
    together = fit_clips_concurrent(clips, DEV, SMALL, seeds=[0, 1, 2], snapshot_interval=10)
    torch.cuda.synchronize()
    print("run", r, " ".join(f"dpsnr {abs(a['psnr_sum']-b['psnr_sum'])/3:.2f} dcount {abs(a['splats_final']-b['splats_final'])/a['splats_final']*100:.1f}%" for a, b in zip(alone, together)), flush=True)
