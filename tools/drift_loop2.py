import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tests.test_gpu_drift import _fit_both, SMALL, DEV
from gflow_amd import synthetic as S
from gflow_amd.fit_video import upload_clip
n = 24
for r in range(5):
  for seed in (0, 1):
    frames = upload_clip(S.make_clip(n, 96, 128, seed=seed, device=DEV), DEV)
    cfg = dict(SMALL, densify_interval=0, densify_interval_after=0)
    (ma, pa, ta), (mb, pb, tb) = _fit_both(frames, cfg)
    d = [x - y for x, y in zip(pa, pb)]
    differ = int((ta.still_mask != tb.still_mask).sum())
    print(f"run {r} seed {seed}: max|d| {max(abs(v) for v in d):.2f} mean {sum(d)/n:+.3f} half2 {sum(d[n//2:])/(n-n//2):+.3f} differ {differ}/{ta.current_pts_num()} ({differ/ta.current_pts_num()*100:.1f}%) "
          f"still {abs(float(ta.still_mask.float().mean()) - float(tb.still_mask.float().mean())):.4f} pose {(ta.pose.detach() - tb.pose.detach()).abs().max().item():.2e}", flush=True)
