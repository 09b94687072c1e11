"""Per-frame PSNR / splat count / estimated camera of a fit of the rigid synthetic clip (analysis tool).
    gpurun -- python tools/rigid_clip_probe.py [frames] [--gt-extr] [--operator] [--h H --w W --n N]
--operator: the reference's loop over the five msplat operators (fused=False) instead of the fused iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
gt = "--gt-extr" in sys.argv
def opt(name, d):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
H, W, N = opt("--h", 480), opt("--w", 854), opt("--n", 60000)
dev = torch.device("cuda", 0)
frames = S.make_clip(n_frames, H, W, seed=0, device=dev)
if gt:
    for f in frames:
        f["extr"] = f["extr_gt"]
frames = FV.upload_clip(frames, dev)
logs = []
import gflow_amd.trainer as TR
poses = []
orig = TR.SimpleGaussian.train_steps
def spy(self, *a, **k):
    r = yield from orig(self, *a, **k)
    poses.append((k.get("camera_only", False), self.pose.detach().cpu().tolist()))
    return r
TR.SimpleGaussian.train_steps = spy
m = FV.fit_clip(frames, dev, dict(num_points=N), seed=0, log=logs.append, fused="--operator" not in sys.argv)
for l in logs:
    print(l)
for i, (cam, p) in enumerate(poses):
    print("train", i, "camera_only" if cam else "joint", " t = %.4f %.4f %.4f  q = %.4f %.4f %.4f %.4f" % (p[4], p[5], p[6], p[0], p[1], p[2], p[3]))
print(m)
