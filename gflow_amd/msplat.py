"""Drop-in for the five ``msplat`` operators GFlow calls, on MI355X.

Same names, positional arguments, shapes and return values as the reference call
sites (gflow/utils/render.py:21-24, 37-41, 44-49, 52-54, 58-64; gflow/trainer.py:955),
each a ``torch.autograd.Function`` over the C ABI of libgflow_hip.so.  Use as

    import gflow_amd.msplat as msplat

``compute_sh`` is provided as an optional sixth operator.

Gradients provided (the ones the reference consumes):
project_point -> xyz, extr; compute_cov3d -> scale, rotate;
ewa_project -> xyz, cov3d, extr; alpha_blending -> uv, conic, opacity, feature.
"""
import torch

from . import _lib as L

NEAREST = 0.2
EXTENT = 1.3
TILE = 16


def _f32(t, name, tail):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"msplat: {name} must be a tensor")
    if t.dtype != torch.float32:
        raise RuntimeError(f"msplat: {name} must be float32, got {t.dtype}")
    if tuple(t.shape[1:]) != tuple(tail) and not (t.dim() == 1 and tail == () ):
        raise RuntimeError(f"msplat: {name} must have shape (N,{','.join(map(str, tail))}), got {tuple(t.shape)}")
    L.need_device(t)
    return t.contiguous()


def _cam(intr, extr):
    if intr.numel() != 4 or extr.numel() != 12:
        raise RuntimeError("msplat: intr must have 4 elements [fx,fy,cx,cy] and extr must be (3,4)")
    L.need_device(intr, extr)
    return intr.detach().float().contiguous(), extr.float().contiguous()


def _vis(visible, n):
    if visible.numel() != n:
        raise RuntimeError("msplat: visible must have N elements")
    L.need_device(visible)
    v = visible.reshape(-1)
    return v.view(torch.uint8) if v.dtype == torch.bool else (v != 0).view(torch.uint8)


def tile_grid(W, H):
    return (W + TILE - 1) // TILE, (H + TILE - 1) // TILE


# ------------------------------------------------------------------ project_point
class _ProjectPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, intr, extr, W, H, nearest, extent):
        lib = L.load()
        n = xyz.shape[0]
        uv = torch.empty((n, 2), dtype=torch.float32, device=xyz.device)
        depth = torch.empty((n, 1), dtype=torch.float32, device=xyz.device)
        L.check(lib.gfl_project_point_fwd(L.ptr(xyz), L.ptr(intr), L.ptr(extr), n, W, H, nearest, extent,
                                          L.ptr(uv), L.ptr(depth), L.stream()), "project_point")
        ctx.save_for_backward(xyz, intr, extr, depth)
        return uv, depth

    @staticmethod
    def backward(ctx, d_uv, d_depth):
        lib = L.load()
        xyz, intr, extr, depth = ctx.saved_tensors
        n = xyz.shape[0]
        d_xyz = torch.empty_like(xyz)
        d_extr = torch.empty((3, 4), dtype=torch.float32, device=xyz.device)
        ws = L.scratch(lib.gfl_reduce_workspace_bytes(n), xyz.device)
        L.check(lib.gfl_project_point_bwd(L.ptr(xyz), L.ptr(intr), L.ptr(extr), L.ptr(depth),
                                          L.ptr(d_uv.contiguous()), L.ptr(d_depth.contiguous()), n, L.ptr(d_xyz),
                                          L.ptr(d_extr), L.ptr(ws), ws.numel(), L.stream()), "project_point backward")
        return d_xyz, None, d_extr, None, None, None, None


def project_point(xyz, intr, extr, W, H, nearest=NEAREST, extent=EXTENT):
    """xyz (N,3), intr (4,), extr (3,4) -> uv (N,2), depth (N,1).  Culled points have
    depth 0 and uv (0,0) (render.py:29 derives ``visible`` from that)."""
    xyz = _f32(xyz, "xyz", (3,))
    intr, extr = _cam(intr, extr)
    return _ProjectPoint.apply(xyz, intr, extr, int(W), int(H), float(nearest), float(extent))


# ------------------------------------------------------------------ compute_cov3d
class _Cov3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, rotate, vis):
        lib = L.load()
        n = scale.shape[0]
        cov = torch.empty((n, 6), dtype=torch.float32, device=scale.device)
        L.check(lib.gfl_cov3d_fwd(L.ptr(scale), L.ptr(rotate), L.ptr(vis), n, L.ptr(cov), L.stream()), "compute_cov3d")
        ctx.save_for_backward(scale, rotate, vis)
        return cov

    @staticmethod
    def backward(ctx, d_cov):
        lib = L.load()
        scale, rotate, vis = ctx.saved_tensors
        n = scale.shape[0]
        d_scale = torch.empty_like(scale)
        d_rot = torch.empty_like(rotate)
        L.check(lib.gfl_cov3d_bwd(L.ptr(scale), L.ptr(rotate), L.ptr(vis), L.ptr(d_cov.contiguous()), n,
                                  L.ptr(d_scale), L.ptr(d_rot), L.stream()), "compute_cov3d backward")
        return d_scale, d_rot, None


def compute_cov3d(scale, rotate, visible):
    """scale (N,3), rotate (N,4) unit WXYZ, visible (N,1) bool -> cov3d (N,6)."""
    scale = _f32(scale, "scale", (3,))
    rotate = _f32(rotate, "rotate", (4,))
    return _Cov3d.apply(scale, rotate, _vis(visible, scale.shape[0]))


# -------------------------------------------------------------------- ewa_project
class _Ewa(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, cov3d, intr, extr, uv, vis, W, H):
        lib = L.load()
        n = xyz.shape[0]
        dev = xyz.device
        conic = torch.empty((n, 3), dtype=torch.float32, device=dev)
        radius = torch.empty((n, 1), dtype=torch.int32, device=dev)
        tiles = torch.empty((n, 1), dtype=torch.int32, device=dev)
        L.check(lib.gfl_ewa_fwd(L.ptr(xyz), L.ptr(cov3d), L.ptr(intr), L.ptr(extr), L.ptr(uv), L.ptr(vis), n, W, H,
                                L.ptr(conic), L.ptr(radius), L.ptr(tiles), L.stream()), "ewa_project")
        ctx.save_for_backward(xyz, cov3d, intr, extr, radius)
        ctx.wh = (W, H)
        ctx.mark_non_differentiable(radius, tiles)
        return conic, radius, tiles

    @staticmethod
    def backward(ctx, d_conic, _r, _t):
        lib = L.load()
        xyz, cov3d, intr, extr, radius = ctx.saved_tensors
        W, H = ctx.wh
        n = xyz.shape[0]
        d_xyz = torch.empty_like(xyz)
        d_cov = torch.empty_like(cov3d)
        d_extr = torch.empty((3, 4), dtype=torch.float32, device=xyz.device)
        ws = L.scratch(lib.gfl_reduce_workspace_bytes(n), xyz.device)
        L.check(lib.gfl_ewa_bwd(L.ptr(xyz), L.ptr(cov3d), L.ptr(intr), L.ptr(extr), L.ptr(radius),
                                L.ptr(d_conic.contiguous()), n, W, H, L.ptr(d_xyz), L.ptr(d_cov), L.ptr(d_extr),
                                L.ptr(ws), ws.numel(), L.stream()), "ewa_project backward")
        return d_xyz, d_cov, None, d_extr, None, None, None, None


def ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible):
    """-> conic (N,3) upper-triangular [a,b,c], radius (N,1) int32, tiles_touched (N,1) int32."""
    xyz = _f32(xyz, "xyz", (3,))
    cov3d = _f32(cov3d, "cov3d", (6,))
    uv = _f32(uv.detach(), "uv", (2,))
    intr, extr = _cam(intr, extr)
    return _Ewa.apply(xyz, cov3d, intr, extr, uv, _vis(visible, xyz.shape[0]), int(W), int(H))


# ------------------------------------------------------------------ sort_gaussian
def sort_gaussian(uv, depth, W, H, radius, tiles_touched):
    """-> gaussian_ids_sorted (K,) int32 ordered by (tile, depth, id), tile_range (T,2)
    int32 [start,end).  Integer outputs, no gradient.  One host read of K (the
    output has a data-dependent size in this API); the fused render path in
    gflow_amd.render avoids it."""
    lib = L.load()
    uv = _f32(uv.detach(), "uv", (2,))
    depth = depth.detach().reshape(-1)
    depth = _f32(depth, "depth", ())
    n = uv.shape[0]
    W, H = int(W), int(H)
    L.need_device(radius)
    radius = radius.reshape(-1).to(torch.int32).contiguous()
    gx, gy = tile_grid(W, H)
    T = gx * gy
    dev = uv.device
    offsets = torch.empty(T + 1, dtype=torch.int32, device=dev)
    L.check(lib.gfl_bin_count(L.ptr(uv), L.ptr(radius), None, n, W, H, L.ptr(offsets), L.stream()), "sort_gaussian")
    K = int(offsets[T].item())
    ids = torch.empty(K, dtype=torch.int32, device=dev)
    tile_range = torch.empty((T, 2), dtype=torch.int32, device=dev)
    overflow = torch.empty(1, dtype=torch.int32, device=dev)
    ws = L.scratch(lib.gfl_bin_workspace_bytes(n, K, W, H), dev)
    L.check(lib.gfl_bin_sort(L.ptr(uv), L.ptr(depth), L.ptr(radius), None, n, W, H, L.ptr(offsets), K, L.ptr(ids),
                             L.ptr(tile_range), L.ptr(overflow), L.ptr(ws), ws.numel(), L.stream()), "sort_gaussian")
    return ids, tile_range


# ----------------------------------------------------------------- alpha_blending
class _Blend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uv, conic, opacity, feature, ids, tile_range, bg, W, H):
        lib = L.load()
        n, C = feature.shape
        dev = feature.device
        out = torch.empty((C, H, W), dtype=torch.float32, device=dev)
        final_T = torch.empty((H, W), dtype=torch.float32, device=dev)
        n_contrib = torch.empty((H, W), dtype=torch.int32, device=dev)
        for c0 in range(0, C, 4):
            cc = min(4, C - c0)
            L.check(lib.gfl_blend_fwd(L.ptr(uv), L.ptr(conic), L.ptr(opacity), L.ptr(feature), C, c0, cc, L.ptr(ids),
                                      L.ptr(tile_range), bg, W, H, L.ptr(out[c0:]), L.ptr(final_T), L.ptr(n_contrib),
                                      L.stream()), "alpha_blending")
        ctx.save_for_backward(uv, conic, opacity, feature, ids, tile_range, final_T, n_contrib)
        ctx.meta = (bg, W, H)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = L.load()
        uv, conic, opacity, feature, ids, tile_range, final_T, n_contrib = ctx.saved_tensors
        bg, W, H = ctx.meta
        n, C = feature.shape
        d_out = d_out.contiguous()
        d_uv = torch.empty_like(uv)
        d_conic = torch.empty_like(conic)
        d_op = torch.empty_like(opacity)
        d_feat = torch.empty_like(feature)
        for c0 in range(0, C, 4):
            cc = min(4, C - c0)
            L.check(lib.gfl_blend_bwd(L.ptr(uv), L.ptr(conic), L.ptr(opacity), L.ptr(feature), C, c0, cc, L.ptr(ids),
                                      L.ptr(tile_range), bg, W, H, L.ptr(final_T), L.ptr(n_contrib),
                                      L.ptr(d_out[c0:]), n, L.ptr(d_uv), L.ptr(d_conic), L.ptr(d_op), L.ptr(d_feat),
                                      1 if c0 == 0 else 0, L.stream()), "alpha_blending backward")
        return d_uv, d_conic, d_op, d_feat, None, None, None, None, None


# The five operators driven one by one in the training pattern (several DIFFERENTIABLE composites over the same sorted lists,
# iteration after iteration: render.py:58-106 under trainer.py:404-407) are the slow level of this drop-in: ~2.2 ms per render
# forward + backward at 480p / 60 k against 0.23-0.35 ms for the fused ``render`` operator (INTEGRATION.md section 1,
# bench.py ``drop_in_levels``).  A maintainer who does only the ``import msplat`` swap gets that silently -- so say it, once.
_TRAINING_PATTERN = {"blends": 0, "warned": False}
_TRAINING_PATTERN_AFTER = 8          # differentiable composites (two iterations of the training call's four)


def _note_training_pattern(*tensors):
    st = _TRAINING_PATTERN
    if st["warned"] or not torch.is_grad_enabled() or not any(t.requires_grad for t in tensors):
        return
    st["blends"] += 1
    if st["blends"] >= _TRAINING_PATTERN_AFTER:
        st["warned"] = True
        import warnings
        warnings.warn(
            "gflow_amd.msplat: the five operators are being driven one by one in a training loop (differentiable "
            "alpha_blending calls).  That is the slow level of this drop-in (about 7-10x the fused path at 480p / 60k splats): "
            "replace utils/render.py's render_multiple by gflow_amd.render.render_multiple (same signature; the training "
            "call's outputs go through ONE fused operator), or call gflow_amd.render.render(gaussians, camera).  "
            "See INTEGRATION.md section 1.", RuntimeWarning, stacklevel=3)


def alpha_blending(uv, conic, opacity, feature, gaussian_ids_sorted, tile_range, bg, W, H):
    """Front-to-back compositing of feature (N,C) -> (C,H,W); ``bg`` is a python
    float applied to every channel (trainer.py:29-36)."""
    _note_training_pattern(uv, conic, opacity, feature)
    uv = _f32(uv, "uv", (2,))
    conic = _f32(conic, "conic", (3,))
    n = uv.shape[0]
    if opacity.numel() != n:
        raise RuntimeError("msplat: opacity must have N elements")
    opacity = _f32(opacity.reshape(n, 1), "opacity", (1,))
    if feature.dim() != 2 or feature.shape[0] != n or feature.shape[1] < 1:
        raise RuntimeError("msplat: feature must be (N,C) with C>=1")
    feature = _f32(feature, "feature", (feature.shape[1],))
    L.need_device(gaussian_ids_sorted, tile_range)
    W, H = int(W), int(H)
    gx, gy = tile_grid(W, H)
    if tile_range.numel() != 2 * gx * gy:
        raise RuntimeError("msplat: tile_range must be (T,2) with T = ceil(W/16)*ceil(H/16)")
    ids = gaussian_ids_sorted.to(torch.int32).contiguous()
    tr = tile_range.to(torch.int32).contiguous()
    return _Blend.apply(uv, conic, opacity, feature, ids, tr, float(bg), W, H)


# --------------------------------------------------------------------- compute_sh
class _ComputeSh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shs, dirs, vis):
        lib = L.load()
        n, k = shs.shape[0], shs.shape[1]
        out = torch.empty((n, 3), dtype=torch.float32, device=shs.device)
        L.check(lib.gfl_sh_fwd(L.ptr(shs), L.ptr(dirs), L.ptr(vis), n, k, L.ptr(out), L.stream()), "compute_sh")
        ctx.save_for_backward(shs, dirs, vis)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = L.load()
        shs, dirs, vis = ctx.saved_tensors
        n, k = shs.shape[0], shs.shape[1]
        d_shs = torch.empty_like(shs)
        d_dirs = torch.empty_like(dirs)
        L.check(lib.gfl_sh_bwd(L.ptr(shs), L.ptr(dirs), L.ptr(vis), L.ptr(d_out.contiguous()), n, k, L.ptr(d_shs),
                               L.ptr(d_dirs), L.stream()), "compute_sh backward")
        return d_shs, d_dirs, None


def compute_sh(shs, view_dirs, visible=None):
    """Optional operator (GFlow never calls it: its colour is sigmoid(rgb), trainer.py:68).
    shs (N,K,3) with K = (degree+1)^2 in {1,4,9,16}, view_dirs (N,3) unit vectors, visible (N,1)
    bool or None -> (N,3) = sum_k Y_k(dir) shs[:,k] in the real SH basis of the 3DGS code base
    (no +0.5 offset, no clamp: callers add them)."""
    if not isinstance(shs, torch.Tensor) or shs.dim() != 3 or shs.shape[2] != 3 or shs.shape[1] not in (1, 4, 9, 16):
        raise RuntimeError("msplat: shs must have shape (N,K,3) with K in {1,4,9,16}")
    if shs.dtype != torch.float32:
        raise RuntimeError(f"msplat: shs must be float32, got {shs.dtype}")
    L.need_device(shs)
    dirs = _f32(view_dirs, "view_dirs", (3,))
    if dirs.shape[0] != shs.shape[0]:
        raise RuntimeError("msplat: view_dirs must have N rows")
    vis = None if visible is None else _vis(visible, shs.shape[0])
    return _ComputeSh.apply(shs.contiguous(), dirs, vis)
