"""Several clips on one GPU: taking turns on every CU (streams) against the chip PARTITIONED between them (CU-masked streams,
each clip on its share of every XCD, grids and queues sized for the share).  Alternating, one box.  (analysis tool)
    gpurun -- python tools/partition_ab.py [frames] [rounds] [levels, e.g. 2,3,4]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
levels = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 3, 4]
dev = torch.device("cuda", 0)
clips = [FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=100 + i, device=dev), dev) for i in range(max(levels))]
cfg = dict(num_points=60000, traj_num=100, traj_offset=2)
FV.fit_clips_concurrent([c[:2] for c in clips[:2]], dev, cfg, snapshot_interval=10)
FV.fit_clips_concurrent([c[:2] for c in clips[:2]], dev, cfg, snapshot_interval=10, partition=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); r1 = FV.fit_clip(clips[0], dev, cfg, seed=0, snapshot_interval=10); torch.cuda.synchronize()
one = time.perf_counter() - t0
print(f"one clip alone: {one:.3f} s = {n_frames / one:.2f} frames/s")
res = {}
for r in range(rounds):
    for c in levels:
        for part in (False, True):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = FV.fit_clips_concurrent(clips[:c], dev, cfg, snapshot_interval=10, partition=part)
            torch.cuda.synchronize(); w = time.perf_counter() - t0
            res.setdefault((c, part), []).append((w, [o["psnr_sum"] for o in out]))
for (c, part), v in sorted(res.items()):
    best = min(w for w, _ in v)
    print(f"{c} clips {'partitioned' if part else 'taking turns'}: best {best:.3f} s = {c * n_frames / best:.2f} frames/s "
          f"({c * n_frames / best / (n_frames / one):.2f}x one clip)   all {[round(w, 3) for w, _ in v]}   psnr sums {[round(p, 3) for p in v[0][1]]}")
