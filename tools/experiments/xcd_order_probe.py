"""Does the ORDER of the splat rows matter?  Workgroup b of a launch runs on XCD b % 8 and every XCD has its own L2: the
per-splat launch (256 rows per workgroup) gathers the pair rows the backward blend wrote from the L2 of the XCD that walks the
splat's tiles (band x of the image -> queues q = x mod 8).  Rows in creation order (random over the image) against rows
interleaved so that block b holds splats of band b % 8:   python tools/experiments/xcd_order_probe.py [block]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gflow_amd import synthetic as S
from gflow_amd.fused import FitEngine, set_profile
from gflow_amd import _lib
import ctypes

H, W, N = 480, 854, 60000
BLOCK = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda"
frame = S.make_frame(H, W, seed=0)
raw = S.init_splats(frame, N, seed=0, grown=True)
NAMES = ("xyz", "scale", "rotate", "opacity", "rgb")


def banded(raw, block):
    # v coordinate of every splat (identity camera): band = eighth of the image height
    f = raw["intr"][1]; cy = raw["intr"][3]
    v = raw["xyz"][:, 1] / raw["xyz"][:, 2] * f + cy
    band = (v / (H / 8.0)).floor().clamp(0, 7).long()
    lists = [torch.nonzero(band == b).flatten() for b in range(8)]
    ptr = [0] * 8
    out = []
    b = 0
    total = sum(len(l) for l in lists)
    while sum(ptr) < total:
        x = b % 8
        if ptr[x] >= len(lists[x]):                       # this band has run out: the longest remaining one
            x = max(range(8), key=lambda k: len(lists[k]) - ptr[k])
        take = lists[x][ptr[x]:ptr[x] + block]
        ptr[x] += len(take)
        out.append(take)
        b += 1
    return torch.cat(out)


def run(tag, perm):
    eng = FitEngine(W, H, capacity=2 * N, device=dev)
    r = {k: (raw[k] if perm is None else raw[k][perm]).to(dev) for k in NAMES}
    eng.set_splats(r)
    eng.intr.copy_(raw["intr"].to(dev))
    eng.set_targets(frame["image"], frame["depth"])
    for k, v in dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=4e-3, lr_camera=0.0, total_iters=500).items():
        setattr(eng.hp, k, v)
    eng.reset_optimizer()
    for _ in range(32):
        eng.iteration()
    torch.cuda.synchronize()
    lib = _lib.load()
    set_profile(0x1ff)
    for _ in range(40):
        eng.iteration()
    torch.cuda.synchronize()
    tot = (ctypes.c_double * 9)(); cnt = (ctypes.c_int * 9)()
    lib.gfl_profile_read(tot, cnt, 9)
    set_profile(0)
    names = ["pre", "colscan", "scatter", "sort", "fwd", "loss", "bwd", "splat", "camera"]
    print(tag, " ".join(f"{n} {1e3 * tot[i] / max(cnt[i], 1):.1f}" for i, n in enumerate(names) if cnt[i]), "K", eng.K)


for rep in range(2):
    run("creation order ", None)
    run(f"banded x{BLOCK}   ", banded(raw, BLOCK))
