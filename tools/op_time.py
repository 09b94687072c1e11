import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import gflow_amd.render as R
from gflow_amd import synthetic as S
DEV = torch.device("cuda", 0)
H, W, N = 480, 854, 60000
frame = S.make_frame(H, W, seed=0)
raw = S.init_splats(frame, N, seed=0, grown=True)
act = dict(xyz=raw["xyz"], scale=raw["scale"].abs(), rotate=torch.nn.functional.normalize(raw["rotate"]),
           opacity=torch.sigmoid(10 * raw["opacity"]), rgb=torch.sigmoid(raw["rgb"]))
leaves = {k: v.to(DEV).requires_grad_(True) for k, v in act.items()}
cam = dict(intr=raw["intr"].to(DEV), extr=raw["extr"].to(DEV), W=W, H=H)
grad = ((torch.rand(3, H, W, device=DEV) - 0.5) / (H * W)).contiguous()
if "--small-first" in sys.argv:
    fs = S.make_frame(48, 64, seed=3)
    rs = S.init_splats(fs, 800, seed=3, grown=True)
    acts = dict(xyz=rs["xyz"], scale=rs["scale"].abs(), rotate=torch.nn.functional.normalize(rs["rotate"]),
                opacity=torch.sigmoid(10 * rs["opacity"]), rgb=torch.sigmoid(rs["rgb"]))
    ls = {k: v.to(DEV).requires_grad_(True) for k, v in acts.items()}
    o = R.render(ls, dict(intr=rs["intr"].to(DEV), extr=rs["extr"].to(DEV), W=64, H=48), 0.0)
    o["rgb"].sum().backward()
    torch.cuda.synchronize()
def fwd():
    with torch.no_grad():
        R.render(leaves, cam, 0.0)
def both():
    out = R.render(leaves, cam, 0.0)
    out["rgb"].backward(grad)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
print("fwd %.3f ms  fwd+bwd %.3f ms" % (timed(fwd) * 1e3, timed(both) * 1e3))
from gflow_amd import _lib
lib = _lib.load()
import bench
lib.gfl_profile_enable(1)
for _ in range(10): both()
torch.cuda.synchronize()
lib.gfl_profile_enable(0)
print({k: round(v * 1e3, 1) for k, v in bench.profile_read(lib).items()})
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): both()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
