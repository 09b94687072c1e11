import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tests.test_gpu_drift import _fit_both, SMALL, DEV
from gflow_amd import synthetic as S
from gflow_amd.fit_video import upload_clip
n = 24
frames = upload_clip(S.make_clip(n, 96, 128, seed=0, device=DEV), DEV)
for r in range(12):
    (ma, pa, ta), (mb, pb, tb) = _fit_both(frames, SMALL)
    d = [x - y for x, y in zip(pa, pb)]
    print(f"run {r}: max|d| {max(abs(v) for v in d):.2f}  mean d {sum(d)/n:+.3f}  counts {ta.current_pts_num()} / {tb.current_pts_num()} ({abs(ta.current_pts_num()-tb.current_pts_num())/tb.current_pts_num()*100:.1f} %)", flush=True)
