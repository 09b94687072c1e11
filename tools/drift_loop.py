"""The statistics the drift tests bound, sampled (tests/test_gpu_drift.py): fused against operator path, run to run.
    python tools/drift_loop.py [runs] [small|full]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_drift import _fit_both, SMALL, DEV
from gflow_amd import synthetic as S
from gflow_amd.fit_video import upload_clip
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
full = len(sys.argv) > 2 and sys.argv[2] == "full"
n = 8 if full else 24
frames = upload_clip(S.make_clip(n, 480, 854, seed=0, device=DEV) if full else S.make_clip(n, 96, 128, seed=0, device=DEV), DEV)
for r in range(runs):
    (ma, pa, ta), (mb, pb, tb) = _fit_both(frames, dict(num_points=60000) if full else SMALL)
    d = [x - y for x, y in zip(pa, pb)]
    print(f"run {r}: max|d| {max(abs(v) for v in d):.2f}  mean d {sum(d)/n:+.3f}  counts {ta.current_pts_num()} / {tb.current_pts_num()} "
          f"({abs(ta.current_pts_num()-tb.current_pts_num())/tb.current_pts_num()*100:.1f} %)  step {pa[0]-pa[1]:.2f} / {pb[0]-pb[1]:.2f}", flush=True)
