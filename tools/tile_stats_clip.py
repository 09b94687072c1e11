"""Tile list lengths after every train() call of a clip fit.   gpurun -- python tools/tile_stats_clip.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gflow_amd import synthetic as S
from gflow_amd import fit_video as FV
from gflow_amd.trainer import SimpleGaussian

frames = S.make_clip(int(sys.argv[1]) if len(sys.argv) > 1 else 3, 480, 854, seed=0)
orig = SimpleGaussian.train
def train(self, *a, **k):
    r = orig(self, *a, **k)
    eng = self.engine
    rng = eng.tile_range.cpu().numpy()
    n = rng[:, 1] - rng[:, 0]
    rad = eng.rec[:eng.N, 11].cpu().numpy().view(np.int32)
    print(f"train(it={k.get('iterations')}, cam={k.get('camera_only', False)}) N {eng.N} K {n.sum()} mean {n.mean():.0f} "
          f"p99 {np.percentile(n, 99):.0f} max {n.max()} >512: {(n > 512).sum()} >1024: {(n > 1024).sum()} >2048: {(n > 2048).sum()} "
          f"rad max {rad.max()} wide(>32 tiles) {(rad > 40).sum()}")
    return r
SimpleGaussian.train = train
FV.fit_clip(frames, torch.device("cuda", 0), dict(num_points=60000), seed=0)
