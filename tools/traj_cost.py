"""Where the per-frame trajectory work of fit_clip goes (traj_num 100): python tools/traj_cost.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV
import gflow_amd.trainer as T

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0), dev)
acc = {}


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    return w


FV.select_traj_seeds = timed("select_traj_seeds", FV.select_traj_seeds)
T.SimpleGaussian.eval_trajectories = timed("eval_trajectories", T.SimpleGaussian.eval_trajectories)
T.SimpleGaussian._render_scene_fused = timed("_render_scene_fused", T.SimpleGaussian._render_scene_fused)
import gflow_amd.hull as Hh
Hh.FastConcaveHull2D.__init__ = timed("hull_init", Hh.FastConcaveHull2D.__init__)
Hh.FastConcaveHull2D.mask = timed("hull_mask", Hh.FastConcaveHull2D.mask)
for traj in (0, 100, 0, 100):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = FV.fit_clip(frames, dev, dict(num_points=60000, traj_num=traj, traj_offset=2), seed=0, snapshot_interval=10)
    torch.cuda.synchronize(); w = time.perf_counter() - t0
    print(f"traj_num {traj}: {w:.4f} s;", {k: (len(v), round(1e3 * sum(v), 2)) for k, v in acc.items()}, "(calls, ms total; the timers synchronise)")
