"""What the blend kernels have to do on the lists a forward left behind (analysis; bench.py calls it once per pinned window):
K pairs, (pixel, splat) pairs with alpha >= 1/255, the (splat, 8x8 block) UNITS the kernels walk -- forward: blocks with a
visible pixel; backward: blocks with a pixel that still needs the splat (at or before its last contributor) -- and the lane
efficiency of a unit (pairs / 64 lanes).  The kernels' own unit test is the exact ellipse-against-box test with a safety
margin (gfl_math.hpp: box_hit), so their counts are a few per cent above "any pixel visible".

    from tools.unit_stats import unit_stats;  unit_stats(engine)       (after a forward; reads the engine's lists on the device)
"""
import torch


def unit_stats(eng, chunk=16384):
    dev = eng.dev
    W, H, gx = eng.W, eng.H, (eng.W + 15) // 16
    rng = eng.tile_range.long()
    # (the lists have gaps between them when the iteration binned into reserved tile regions: gathered by tile range)
    lens = rng[:, 1] - rng[:, 0]
    K = int(lens.sum())
    if K == 0:
        return {"K": 0, "splats": int(eng.N)}
    tile_of = torch.repeat_interleave(torch.arange(eng.T, device=dev), lens)
    pos = torch.arange(K, device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
    ids = eng.ids[rng[tile_of, 0] + pos].long()
    rec = eng.rec[ids]
    u, v, A, B, C, o = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5]
    tx, ty = (tile_of % gx) * 16, (tile_of // gx) * 16
    ncontrib = eng.n_contrib.long()
    yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
    tot = dict(valid=0, live=0, u_fwd=0, u_bwd=0)
    edges = torch.tensor([0, 1, 2, 4, 8, 16, 32, 64], device=dev)          # buckets (0,1], (1,2], (2,4], ... (32,64] lanes of a unit
    h_fwd = torch.zeros(7, dtype=torch.long, device=dev)
    h_bwd = torch.zeros(7, dtype=torch.long, device=dev)
    for s in range(0, K, chunk):
        sl = slice(s, min(K, s + chunk))
        fx = (tx[sl, None, None] + xx[None]).float()
        fy = (ty[sl, None, None] + yy[None]).float()
        inside = (fx < W) & (fy < H)
        dx, dy = u[sl, None, None] - fx, v[sl, None, None] - fy
        power = -0.5 * (A[sl, None, None] * dx * dx + C[sl, None, None] * dy * dy) - B[sl, None, None] * dx * dy
        alpha = torch.clamp(o[sl, None, None] * torch.exp(torch.clamp(power, max=0)), max=0.99)
        valid = (power <= 0) & (alpha >= 1 / 255) & inside
        px = torch.clamp(fx.long(), max=W - 1)
        py = torch.clamp(fy.long(), max=H - 1)
        live = valid & (pos[sl, None, None] < ncontrib[py, px])
        v8 = valid.view(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64)
        l8 = live.view(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64)
        tot["valid"] += int(valid.sum())
        tot["live"] += int(live.sum())
        tot["u_fwd"] += int(v8.any(-1).sum())
        tot["u_bwd"] += int(l8.any(-1).sum())
        nv, nl = v8.sum(-1).flatten(), l8.sum(-1).flatten()
        h_fwd += torch.histc(torch.bucketize(nv[nv > 0], edges, right=False).float() - 1, bins=7, min=0, max=7).long()
        h_bwd += torch.histc(torch.bucketize(nl[nl > 0], edges, right=False).float() - 1, bins=7, min=0, max=7).long()
    ln = lens.float()
    return {"K": K, "splats": int(eng.N), "pixel_splat_pairs_visible": tot["valid"], "pixel_splat_pairs_live": tot["live"],
            "units_8x8_fwd": tot["u_fwd"], "units_8x8_bwd": tot["u_bwd"],
            "lane_efficiency_fwd": tot["valid"] / max(64 * tot["u_fwd"], 1),
            "lane_efficiency_bwd": tot["live"] / max(64 * tot["u_bwd"], 1),
            # share of the units by how many of their 64 lanes see the splat: 1, 2, 3-4, 5-8, 9-16, 17-32, 33-64
            "lanes_per_unit_hist_fwd": [round(float(v) / max(tot["u_fwd"], 1), 4) for v in h_fwd.tolist()],
            "lanes_per_unit_hist_bwd": [round(float(v) / max(tot["u_bwd"], 1), 4) for v in h_bwd.tolist()],
            "tile_list_mean": float(ln.mean()), "tile_list_p99": float(ln.kthvalue(max(1, int(0.99 * ln.numel()))).values),
            "tile_list_max": int(lens.max()), "tiles_over_512": int((lens > 512).sum())}
