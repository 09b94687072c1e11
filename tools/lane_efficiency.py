"""How many of the lanes the blend kernels spend on a (splat, 8x8 block) unit see the splat at all?  (analysis tool)
Counts, over the tile lists of the bench scene and of a real first-frame fit, the pixels with alpha >= 1/255 per unit
and what coarser/finer unit shapes or an exact ellipse test would evaluate.
    gpurun -- python tools/lane_efficiency.py [--fit] [--json out.json]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gflow_amd import synthetic as S
from gflow_amd.trainer import SimpleGaussian

dev = torch.device("cuda", 0)
REAL = "--fit" in sys.argv
if REAL:
    from gflow_amd.fit_video import DEFAULTS as c
    frame = S.make_clip(1, bench.H, bench.W, seed=0)[0]
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    tr.init_gaussians_from_image(frame["image"], frame["depth"], num_points=bench.N_SPLATS)
    tr.train(iterations=c["iterations_first"], lr=c["lr"], lr_camera=c["lr_camera"], lambda_var=c["lambda_var"],
             lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"], densify_interval=c["densify_interval"],
             densify_times=c["densify_times"], move_mask=frame["move_mask"])
    for _ in range(3):
        tr.engine.iteration()
else:
    frame = S.make_frame(bench.H, bench.W, seed=0)
    raw = S.init_splats(frame, bench.N_SPLATS, seed=0, grown=True)
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
        tr._attributes[k] = raw[k].to(dev)
    stepper = tr.make_stepper(iterations=500, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
                              move_mask=frame["move_mask"], densify_interval=0, snapshot_interval=0)
    # (the middle of bench.py's pinned window, iterations [12, 32) of this fit)
    for _ in range(bench.STEP_I0 + bench.STEP_WINDOW // 2):
        stepper()
torch.cuda.synchronize()
eng = tr.engine
W, H, gx = eng.W, eng.H, (eng.W + 15) // 16
rng = eng.tile_range.long()
# (the lists have gaps between them when the iteration binned into reserved tile regions: gathered by tile range)
lens = rng[:, 1] - rng[:, 0]
K = int(lens.sum())
tile_of = torch.repeat_interleave(torch.arange(eng.T, device=dev), lens)
pos = torch.arange(K, device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
ids = eng.ids[rng[tile_of, 0] + pos].long()
rec = eng.rec[ids]                                                   # [K, 12]
u, v, A, B, C, o, cutoff = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5], rec[:, 10]
tx, ty = (tile_of % gx) * 16, (tile_of // gx) * 16
ncontrib = eng.n_contrib.long()

def box_hit(u, v, cutoff, x_lo, x_hi, y_lo, y_hi):
    ddx = torch.clamp(torch.maximum(x_lo - u[:, None], u[:, None] - x_hi), min=0)
    ddy = torch.clamp(torch.maximum(y_lo - v[:, None], v[:, None] - y_hi), min=0)
    return ddx * ddx + ddy * ddy <= cutoff[:, None]

q_hits = torch.zeros(eng.T, 16, dtype=torch.long, device=dev)      # live (splat, 4x4 quarter) units per tile and quarter
b_hits = torch.zeros(eng.T, 4, dtype=torch.long, device=dev)       # live (splat, 8x8 block) units per tile and block
tot = dict(valid=0, valid_live=0, u8=0, u8_any=0, u8_any_live=0, u4=0, u4_any=0, u4_any_live=0, rows8=0, rows8_live=0,
           u8x4=0, u8x4_any=0)
CH = 16384
yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
for s in range(0, K, CH):
    e = min(K, s + CH)
    sl = slice(s, e)
    fx = (tx[sl, None, None] + xx[None]).float()
    fy = (ty[sl, None, None] + yy[None]).float()
    inside = (fx < W) & (fy < H)
    dx, dy = u[sl, None, None] - fx, v[sl, None, None] - fy
    power = -0.5 * (A[sl, None, None] * dx * dx + C[sl, None, None] * dy * dy) - B[sl, None, None] * dx * dy
    alpha = torch.clamp(o[sl, None, None] * torch.exp(torch.clamp(power, max=0)), max=0.99)
    valid = (power <= 0) & (alpha >= 1 / 255) & inside
    px = torch.clamp(fx.long(), max=W - 1); py = torch.clamp(fy.long(), max=H - 1)
    live = valid & (pos[sl, None, None] < ncontrib[py, px])           # what the backward pass still needs
    tot["valid"] += int(valid.sum()); tot["valid_live"] += int(live.sum())
    # 8x8 blocks (the current unit), by the disc test of block_mask
    b8 = torch.arange(4, device=dev)
    x_lo = (tx[sl, None] + (b8[None] & 1) * 8).float(); y_lo = (ty[sl, None] + (b8[None] >> 1) * 8).float()
    hit8 = box_hit(u[sl], v[sl], cutoff[sl], x_lo, x_lo + 7, y_lo, y_lo + 7)
    v8 = valid.view(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64)
    l8 = live.view(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64)
    tot["u8"] += int(hit8.sum()); tot["u8_any"] += int(v8.any(-1).sum()); tot["u8_any_live"] += int(l8.any(-1).sum())
    tot["rows8"] += int(v8.view(-1, 4, 8, 8).any(-1).sum()); tot["rows8_live"] += int(l8.view(-1, 4, 8, 8).any(-1).sum())
    # 8 wide x 4 high half blocks
    v84 = valid.view(-1, 4, 4, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 8, 32)
    tot["u8x4_any"] += int(v84.any(-1).sum())
    # 4x4 sub-blocks, disc test
    b4 = torch.arange(16, device=dev)
    x_lo = (tx[sl, None] + (b4[None] & 3) * 4).float(); y_lo = (ty[sl, None] + (b4[None] >> 2) * 4).float()
    hit4 = box_hit(u[sl], v[sl], cutoff[sl], x_lo, x_lo + 3, y_lo, y_lo + 3)
    v4 = valid.view(-1, 4, 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 16, 16)
    l4 = live.view(-1, 4, 4, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 16, 16)
    tot["u4"] += int(hit4.sum()); tot["u4_any"] += int(v4.any(-1).sum()); tot["u4_any_live"] += int(l4.any(-1).sum())
    q_hits.index_add_(0, tile_of[sl], l4.any(-1).long())
    b_hits.index_add_(0, tile_of[sl], l8.any(-1).long())

print("scene:", "real first-frame fit" if REAL else "bench scene", " N", eng.N, " K", K)
print("pixel-splat pairs with alpha >= 1/255: %d  (still needed by the backward pass: %d)" % (tot["valid"], tot["valid_live"]))
def line(name, lanes):
    print("  %-58s %9d lanes  efficiency %.3f (live %.3f)" % (name, lanes, tot["valid"] / lanes, tot["valid_live"] / lanes))
line("8x8 units by the disc test (current)", tot["u8"] * 64)
line("8x8 units with any visible pixel (exact test)", tot["u8_any"] * 64)
line("8x8 units with any LIVE pixel", tot["u8_any_live"] * 64)
line("8x4 half units with any visible pixel", tot["u8x4_any"] * 32)
line("4x4 units by the disc test", tot["u4"] * 16)
line("4x4 units with any visible pixel", tot["u4_any"] * 16)
line("4x4 units with any LIVE pixel", tot["u4_any_live"] * 16)
line("8x1 rows with any visible pixel", tot["rows8"] * 8)
line("8x1 rows with any LIVE pixel", tot["rows8_live"] * 8)
# a wave = an 8x8 block whose four 16-lane rows walk the block's four 4x4 quarters, each its own hit list, in lockstep:
# steps = the longest of the four lists (quarter q of block b: (qy, qx) = (2 (b >> 1) + (q >> 1), 2 (b & 1) + (q & 1)))
qh = q_hits.view(-1, 4, 4)                       # [tile][qy][qx]
per_block = torch.stack([qh[:, 2 * (b >> 1):2 * (b >> 1) + 2, 2 * (b & 1):2 * (b & 1) + 2].reshape(-1, 4) for b in range(4)], dim=1)   # [tile][block][4 quarters]
steps_q = int(per_block.max(-1).values.sum())
print("backward, one splat per step and 8x8 block (shipped): %d steps;  four quarter rows in lockstep: %d steps (%.2fx), "
      "their lists hold %d units (%.2f of a step's four slots used)" % (int(b_hits.sum()), steps_q, steps_q / max(int(b_hits.sum()), 1),
                                                                         int(per_block.sum()), int(per_block.sum()) / max(4 * steps_q, 1)))

if "--json" in sys.argv:
    import json
    # the kernels' unit is an 8x8 block that the exact ellipse test lets through (forward: any visible pixel; backward: any
    # pixel that still needs the splat)
    json.dump({"blend_fwd": tot["valid"] / (tot["u8_any"] * 64), "blend_bwd": tot["valid_live"] / (tot["u8_any_live"] * 64),
               "visible_pixel_splat_pairs": tot["valid"], "live_pairs_backward": tot["valid_live"],
               "units_8x8_forward": tot["u8_any"], "units_8x8_backward": tot["u8_any_live"], "K": K,
               "scene": "bench scene at iteration %d (middle of the pinned window)" % (bench.STEP_I0 + bench.STEP_WINDOW // 2)},
              open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
