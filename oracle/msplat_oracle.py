"""Eager-PyTorch CPU restatement of the five rasteriser operators GFlow calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED against real
msplat: its source is not in /root/reference.  What IS followed, line by line, is
how the reference calls and composes the operators:

  project_point   gflow/utils/render.py:21-24, gflow/trainer.py:955
  compute_cov3d   gflow/utils/render.py:37-41
  ewa_project     gflow/utils/render.py:44-49
  sort_gaussian   gflow/utils/render.py:52-54
  alpha_blending  gflow/utils/render.py:58-64,68-74,84-90,99-105

Everything inside the operators follows the published 3DGS / EWA formulation; the
constants that are internal to msplat (and therefore assumptions)
are the module-level names below and mirror include/gflow_hip.h.

All differentiable operators are written with plain torch ops so that
``torch.autograd`` provides the reference gradients (float32 or float64).
"""
import math

import numpy as np
import torch

TILE = 16                 # tile edge in pixels
NEAREST = 0.2             # near-plane cull: visible iff z_cam > NEAREST
EXTENT = 1.3              # frustum margin: |u - W/2| <= EXTENT * W/2 (same for v)
FOV_CLAMP = 1.3           # clamp of x/z, y/z in the EWA Jacobian (3DGS)
LOWPASS = 0.3             # added to the diagonal of the 2-D covariance
EIG_FLOOR = 0.1           # floor under the eigenvalue discriminant
RADIUS_SIGMA = 3.0        # pixel radius = ceil(RADIUS_SIGMA * sqrt(lambda_max))
ALPHA_MIN = 1.0 / 255.0   # skip a splat at a pixel when alpha < ALPHA_MIN
ALPHA_MAX = 0.99          # alpha cap (straight-through in the backward, 3DGS)
T_MIN = 1e-4              # stop compositing a pixel when T would drop below
PIXEL_CENTER = 0.0        # pixel (x, y) is sampled at (x + PIXEL_CENTER, y + PIXEL_CENTER): 0 = 3DGS, 0.5 = gsplat


# --------------------------------------------------------------------------- A4
def project_point(xyz, intr, extr, W, H, nearest=NEAREST, extent=EXTENT):
    """xyz (N,3), intr (4,) [fx,fy,cx,cy], extr (3,4) world->camera.

    Returns uv (N,2), depth (N,1).  Culled points give uv = (0,0), depth = 0,
    which is what gflow/utils/render.py:29 (``visible = depth != 0``) and
    gflow/trainer.py:424 (``uv > 0`` tests) rely on.
    """
    fx, fy, cx, cy = intr[0], intr[1], intr[2], intr[3]
    R = extr[:, :3]
    t = extr[:, 3]
    pc = xyz @ R.T + t
    z = pc[:, 2]
    front = z > nearest
    zs = torch.where(front, z, torch.ones_like(z))
    u = fx * pc[:, 0] / zs + cx
    v = fy * pc[:, 1] / zs + cy
    lo_u, hi_u = (1.0 - extent) * 0.5 * W, (1.0 + extent) * 0.5 * W
    lo_v, hi_v = (1.0 - extent) * 0.5 * H, (1.0 + extent) * 0.5 * H
    vis = front & (u >= lo_u) & (u <= hi_u) & (v >= lo_v) & (v <= hi_v)
    zero = torch.zeros_like(u)
    uv = torch.stack([torch.where(vis, u, zero), torch.where(vis, v, zero)], dim=1)
    depth = torch.where(vis, z, zero).unsqueeze(1)
    return uv, depth


# --------------------------------------------------------------------------- A5
def quat_to_rotmat(q):
    """q (N,4) in WXYZ order (gflow/trainer.py:723,748,932 write identity as
    [1,0,0,0]); not re-normalised here (the caller passes F.normalize output,
    gflow/trainer.py:66)."""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r00 = 1 - 2 * (y * y + z * z)
    r01 = 2 * (x * y - w * z)
    r02 = 2 * (x * z + w * y)
    r10 = 2 * (x * y + w * z)
    r11 = 1 - 2 * (x * x + z * z)
    r12 = 2 * (y * z - w * x)
    r20 = 2 * (x * z - w * y)
    r21 = 2 * (y * z + w * x)
    r22 = 1 - 2 * (x * x + y * y)
    return torch.stack([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).reshape(-1, 3, 3)


def compute_cov3d(scale, rotate, visible):
    """Sigma = R diag(s^2) R^T, six unique entries [xx,xy,xz,yy,yz,zz]; rows of
    invisible splats are zero."""
    Rm = quat_to_rotmat(rotate)
    M = Rm * scale.unsqueeze(1)          # R @ diag(s)
    S = M @ M.transpose(1, 2)
    cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)
    vis = visible.reshape(-1, 1).to(torch.bool)
    return torch.where(vis, cov, torch.zeros_like(cov))


# --------------------------------------------------------------------------- A6
def tile_grid(W, H):
    return (W + TILE - 1) // TILE, (H + TILE - 1) // TILE


def tile_rect(uv, radius, W, H):
    """Integer tile rectangle [x0,x1) x [y0,y1) a splat of pixel radius r covers
    (3DGS getRect: C-style float->int truncation, then clamp to the grid)."""
    gx, gy = tile_grid(W, H)
    r = radius.reshape(-1).to(uv.dtype)
    u, v = uv[:, 0].detach(), uv[:, 1].detach()
    x0 = torch.trunc((u - r) / TILE).clamp(0, gx).to(torch.int64)
    x1 = torch.trunc((u + r + (TILE - 1)) / TILE).clamp(0, gx).to(torch.int64)
    y0 = torch.trunc((v - r) / TILE).clamp(0, gy).to(torch.int64)
    y1 = torch.trunc((v + r + (TILE - 1)) / TILE).clamp(0, gy).to(torch.int64)
    return x0, x1, y0, y1


def ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible):
    """Sigma2 = J W Sigma W^T J^T + LOWPASS*I; conic = Sigma2^-1 as upper-triangular
    [a,b,c] (gflow/utils/render.py:95-96 writes the identity as [1,0,1]);
    radius (N,1) int32; tiles_touched (N,1) int32."""
    fx, fy = intr[0], intr[1]
    R = extr[:, :3]
    t = extr[:, 3]
    vis = visible.reshape(-1).to(torch.bool)
    pc = xyz @ R.T + t
    z = torch.where(vis, pc[:, 2], torch.ones_like(pc[:, 2]))
    limx = FOV_CLAMP * W / (2.0 * fx)
    limy = FOV_CLAMP * H / (2.0 * fy)
    tx = torch.maximum(torch.minimum(pc[:, 0] / z, limx), -limx) * z
    ty = torch.maximum(torch.minimum(pc[:, 1] / z, limy), -limy) * z
    j00 = fx / z
    j02 = -fx * tx / (z * z)
    j11 = fy / z
    j12 = -fy * ty / (z * z)
    m0 = j00.unsqueeze(1) * R[0] + j02.unsqueeze(1) * R[2]
    m1 = j11.unsqueeze(1) * R[1] + j12.unsqueeze(1) * R[2]
    sxx, sxy, sxz, syy, syz, szz = [cov3d[:, i] for i in range(6)]

    def sig(vv):
        return torch.stack([sxx * vv[:, 0] + sxy * vv[:, 1] + sxz * vv[:, 2],
                            sxy * vv[:, 0] + syy * vv[:, 1] + syz * vv[:, 2],
                            sxz * vv[:, 0] + syz * vv[:, 1] + szz * vv[:, 2]], dim=1)

    s0, s1 = sig(m0), sig(m1)
    a = (m0 * s0).sum(1) + LOWPASS
    b = (m0 * s1).sum(1)
    c = (m1 * s1).sum(1) + LOWPASS
    det = a * c - b * b
    ok = vis & (det != 0)
    dets = torch.where(ok, det, torch.ones_like(det))
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=EIG_FLOOR))
    radius = torch.ceil(RADIUS_SIGMA * torch.sqrt(lam.detach()))
    radius = torch.where(ok, radius, torch.zeros_like(radius))
    x0, x1, y0, y1 = tile_rect(uv, radius, W, H)
    tiles = (x1 - x0) * (y1 - y0)
    tiles = torch.where(ok, tiles, torch.zeros_like(tiles))
    ok = ok & (tiles > 0)
    zero = torch.zeros_like(a)
    conic = torch.stack([torch.where(ok, c / dets, zero),
                         torch.where(ok, -b / dets, zero),
                         torch.where(ok, a / dets, zero)], dim=1)
    radius = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32).unsqueeze(1)
    tiles = tiles.to(torch.int32).unsqueeze(1)
    return conic, radius, tiles


# --------------------------------------------------------------------------- A7
def sort_gaussian(uv, depth, W, H, radius, tiles_touched):
    """Duplicate each splat per covered tile, order by (tile, depth, splat id).

    Returns gaussian_ids_sorted (K,) int32 and tile_range (T,2) int32 [start,end).
    Equal depths inside one tile are ordered by splat id (what a stable radix sort
    of (tile<<32 | depth-bits) keys emitted in id order gives)."""
    gx, gy = tile_grid(W, H)
    x0, x1, y0, y1 = [a.numpy() for a in tile_rect(uv.detach().float(), radius, W, H)]
    r = radius.reshape(-1).numpy()
    live = r > 0
    nx = np.where(live, x1 - x0, 0)
    ny = np.where(live, y1 - y0, 0)
    cnt = nx * ny
    K = int(cnt.sum())
    T = gx * gy
    tile_range = np.zeros((T, 2), dtype=np.int32)
    if K == 0:
        return torch.zeros((0,), dtype=torch.int32), torch.from_numpy(tile_range)
    gid = np.repeat(np.arange(len(cnt)), cnt)
    start = np.cumsum(cnt) - cnt
    local = np.arange(K) - np.repeat(start, cnt)
    nxr = np.repeat(nx, cnt)
    ty = np.repeat(y0, cnt) + local // nxr
    tx = np.repeat(x0, cnt) + local % nxr
    tile = ty * gx + tx
    dbits = depth.detach().float().reshape(-1).numpy().view(np.uint32)[gid]
    order = np.lexsort((gid, dbits, tile))
    ids = gid[order].astype(np.int32)
    tile_sorted = tile[order]
    counts = np.bincount(tile_sorted, minlength=T)
    ends = np.cumsum(counts)
    starts = ends - counts
    nonempty = counts > 0
    tile_range[nonempty, 0] = starts[nonempty]
    tile_range[nonempty, 1] = ends[nonempty]
    return torch.from_numpy(ids), torch.from_numpy(tile_range)


# --------------------------------------------------------------------------- A8
def _straight_through_min(x, cap):
    """min(x, cap) in the forward, identity in the backward: 3DGS does not cut
    the gradient at the alpha cap."""
    return x + (torch.clamp(x, max=cap) - x).detach()


def alpha_blending(uv, conic, opacity, feature, gaussian_ids_sorted, tile_range, bg, W, H,
                   max_elems=6_000_000):
    """Front-to-back compositing.  feature (N,C) -> out (C,H,W).

    Per pixel (x,y), sampled at p = (x + PIXEL_CENTER, y + PIXEL_CENTER), for the splats of its tile in sorted
    order: d = uv_i - p; power = -0.5(a dx^2 + c dy^2) - b dx dy; skip if power > 0;
    alpha = min(ALPHA_MAX, o_i exp(power)); skip if alpha < ALPHA_MIN; stop (before
    adding this splat) when T(1-alpha) < T_MIN; out += f_i alpha T; T *= 1-alpha.
    Finally out += T * bg (bg is a python float applied to every channel,
    gflow/trainer.py:29-36)."""
    C = feature.shape[1]
    dt = feature.dtype
    gx, gy = tile_grid(W, H)
    T = gx * gy
    tr = tile_range.to(torch.int64)
    lens = (tr[:, 1] - tr[:, 0])
    out = torch.full((C, gy * TILE, gx * TILE), float(bg), dtype=dt)
    ids_all = gaussian_ids_sorted.to(torch.int64)
    if ids_all.numel() > 0:
        order = torch.argsort(lens, descending=True)
        order = order[lens[order] > 0]
        px_off = torch.arange(TILE, dtype=dt) + PIXEL_CENTER
        pos = 0
        pieces = []
        while pos < order.numel():
            L = int(lens[order[pos]])
            nt = max(1, min(order.numel() - pos, max_elems // (TILE * TILE * L)))
            tids = order[pos:pos + nt]
            pos += nt
            ar = torch.arange(L)
            idx = tr[tids, 0].unsqueeze(1) + ar.unsqueeze(0)                # (nt,L)
            valid = ar.unsqueeze(0) < lens[tids].unsqueeze(1)                # (nt,L)
            idx = torch.where(valid, idx, torch.zeros_like(idx))
            g = ids_all[idx]                                                 # (nt,L)
            tx = (tids % gx).to(dt) * TILE
            ty = (tids // gx).to(dt) * TILE
            pxx = (tx.unsqueeze(1) + px_off.unsqueeze(0))                    # (nt,16)
            pyy = (ty.unsqueeze(1) + px_off.unsqueeze(0))
            # pixel grid (nt,256): row-major inside the tile
            PX = pxx.unsqueeze(1).expand(nt, TILE, TILE).reshape(nt, -1, 1)
            PY = pyy.unsqueeze(2).expand(nt, TILE, TILE).reshape(nt, -1, 1)
            gu = uv[g]                                                       # (nt,L,2)
            gc = conic[g]
            go = opacity[g].reshape(nt, 1, L)
            gf = feature[g]                                                  # (nt,L,C)
            dx = gu[:, :, 0].unsqueeze(1) - PX                               # (nt,256,L)
            dy = gu[:, :, 1].unsqueeze(1) - PY
            power = -0.5 * (gc[:, :, 0].unsqueeze(1) * dx * dx + gc[:, :, 2].unsqueeze(1) * dy * dy) \
                - gc[:, :, 1].unsqueeze(1) * dx * dy
            araw = go * torch.exp(power)
            alpha = _straight_through_min(araw, ALPHA_MAX)
            use = valid.unsqueeze(1) & (power <= 0) & (alpha.detach() >= ALPHA_MIN)
            alpha = torch.where(use, alpha, torch.zeros_like(alpha))
            one_m = 1.0 - alpha
            incl = torch.cumprod(one_m, dim=2)
            contrib = use & (incl.detach() >= T_MIN)
            # drop everything from the first terminated splat on
            one_m = torch.where(contrib, one_m, torch.ones_like(one_m))
            alpha = torch.where(contrib, alpha, torch.zeros_like(alpha))
            incl = torch.cumprod(one_m, dim=2)
            excl = torch.cat([torch.ones_like(incl[:, :, :1]), incl[:, :, :-1]], dim=2)
            wgt = alpha * excl                                               # (nt,256,L)
            col = torch.einsum('tpl,tlc->tcp', wgt, gf)                      # (nt,C,256)
            col = col + incl[:, :, -1].unsqueeze(1) * float(bg)
            pieces.append((tids, col))
        tid_cat = torch.cat([p[0] for p in pieces])
        col_cat = torch.cat([p[1] for p in pieces])                          # (n,C,256)
        full = torch.full((T, C, TILE * TILE), float(bg), dtype=dt)
        full = full.index_copy(0, tid_cat, col_cat)
        out = full.reshape(gy, gx, C, TILE, TILE).permute(2, 0, 3, 1, 4).reshape(C, gy * TILE, gx * TILE)
    return out[:, :H, :W].contiguous()


# ------------------------------------------------------------ A3 render_multiple
def turbo_lut():
    """256x3 turbo table; loaded from the golden fixture captured from matplotlib
    through the reference's apply_float_colormap (tests/golden/make_golden.py)."""
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "gflow_amd", "data", "colormaps.npz")
    return torch.from_numpy(np.load(p)["turbo"]).float()


def apply_float_colormap(image, lut, non_zero=False):
    """gflow/utils/color.py:24-44 restated without the host round trip."""
    if non_zero:
        image = image - torch.min(image[image != 0])
    else:
        image = image - torch.min(image)
    image = image / (torch.max(image) + 1e-5)
    image = torch.nan_to_num(torch.clip(image, 0, 1), 0)
    idx = (image * 255).long()[..., 0]
    return lut.to(image.dtype)[idx]


def render_multiple(input_group, return_type=("rgb", "uv", "depth", "depth_map", "depth_map_color", "center")):
    """gflow/utils/render.py:6-108 with the oracle operators."""
    xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H = input_group
    out = {}
    uv, depth = project_point(xyz, intr, extr, W, H)
    visible = depth != 0
    if "uv" in return_type:
        out["uv"] = uv
    if "depth" in return_type:
        out["depth"] = depth
    cov3d = compute_cov3d(scale, rotate, visible)
    conic, radius, tiles = ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
    ids, tile_range = sort_gaussian(uv, depth, W, H, radius, tiles)
    if "rgb" in return_type:
        out["rgb"] = alpha_blending(uv, conic, opacity, rgb, ids, tile_range, bg, W, H)
    if "depth_map" in return_type:
        out["depth_map"] = alpha_blending(uv, conic, opacity, depth, ids, tile_range, bg, W, H)
    if "depth_map_color" in return_type:
        dc = apply_float_colormap(depth.detach(), turbo_lut(), non_zero=True)
        out["depth_map_color"] = alpha_blending(uv, conic, opacity, dc, ids, tile_range, bg, W, H)
    if "center" in return_type:
        conic1 = torch.ones_like(conic) * torch.tensor([1.0, 0.0, 1.0], dtype=conic.dtype)
        out["center"] = alpha_blending(uv, conic1, torch.ones_like(opacity), rgb, ids, tile_range, bg, W, H)
    return out


def render_traj(input_group, point_num, line_scale=1.0, point_scale=2.0):
    """gflow/utils/render.py:110-156: every splat drawn as an isotropic blob -- conic [1,0,1] * line_scale for the
    last ``point_num`` rows, * point_scale for the others (:144-146) -- over the lists of the real footprints."""
    xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H = input_group
    uv, depth = project_point(xyz, intr, extr, W, H)
    visible = depth != 0
    cov3d = compute_cov3d(scale, rotate, visible)
    conic, radius, tiles = ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
    ids, tile_range = sort_gaussian(uv, depth, W, H, radius, tiles)
    unit = torch.tensor([1.0, 0.0, 1.0], dtype=conic.dtype)
    conic = torch.ones_like(conic) * unit * line_scale
    conic[:-point_num] = torch.ones_like(conic[:-point_num]) * unit * point_scale
    return alpha_blending(uv, conic, opacity, rgb, ids, tile_range, bg, W, H)


def alpha_blending_loops(uv, conic, opacity, feature, gaussian_ids_sorted, tile_range, bg, W, H):
    """Literal per-pixel restatement of the compositing loop (pure python, small
    cases only); an independent check on the vectorised alpha_blending above.
    Also returns final_T (H,W) and n_contrib (H,W) = 1 + list position of the last
    contributing splat."""
    C = feature.shape[1]
    gx, _ = tile_grid(W, H)
    out = np.zeros((C, H, W), dtype=np.float64)
    final_T = np.ones((H, W), dtype=np.float64)
    n_contrib = np.zeros((H, W), dtype=np.int32)
    uvn, cn, on, fn = (a.detach().double().numpy() for a in (uv, conic, opacity.reshape(-1), feature))
    ids = gaussian_ids_sorted.numpy()
    tr = tile_range.numpy()
    for y in range(H):
        for x in range(W):
            s, e = tr[(y // TILE) * gx + (x // TILE)]
            T = 1.0
            acc = np.zeros(C)
            last = 0
            for k in range(s, e):
                g = ids[k]
                dx, dy = uvn[g, 0] - (x + PIXEL_CENTER), uvn[g, 1] - (y + PIXEL_CENTER)
                power = -0.5 * (cn[g, 0] * dx * dx + cn[g, 2] * dy * dy) - cn[g, 1] * dx * dy
                if power > 0:
                    continue
                alpha = min(ALPHA_MAX, on[g] * math.exp(power))
                if alpha < ALPHA_MIN:
                    continue
                if T * (1 - alpha) < T_MIN:
                    break
                acc += fn[g] * alpha * T
                T *= 1 - alpha
                last = k - s + 1
            out[:, y, x] = acc + T * bg
            final_T[y, x] = T
            n_contrib[y, x] = last
    return out, final_T, n_contrib


# --------------------------------------------------------------------------- A17
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def sh_basis(dirs, K):
    """Real SH basis values (N,K) at unit directions (N,3), degrees 0..3, in the sign convention of
    the 3D Gaussian Splatting code base.  No reference call site exists (GFlow never evaluates SH);
    pinned by orthonormality over the sphere in tests/test_oracle_kat.py."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    b = [torch.full_like(x, SH_C0)]
    if K >= 4:
        b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if K >= 9:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if K >= 16:
        b += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
              SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
              SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b[:K], dim=1)


def compute_sh(shs, view_dirs, visible=None):
    """shs (N,K,3), view_dirs (N,3) -> (N,3); rows with visible == False are zero."""
    out = (sh_basis(view_dirs, shs.shape[1]).unsqueeze(2) * shs).sum(dim=1)
    if visible is not None:
        out = out * visible.reshape(-1, 1).to(out.dtype)
    return out
