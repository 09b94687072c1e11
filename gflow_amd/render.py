"""Render orchestrator -- the counterpart of gflow/utils/render.py.

``render_multiple`` / ``render_traj`` / ``render2img`` keep the reference's
signatures and semantics (render.py:6-166) on top of ``gflow_amd.msplat``.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _lib as L
from . import msplat
from .color import apply_float_colormap

FUSED_TYPES = frozenset(("rgb", "uv", "depth", "depth_map"))
USE_FUSED = True        # render_multiple routes {"rgb","uv","depth","depth_map"} requests through the fused operator

# ------------------------------------------------------------ fused differentiable operator
_POOL = {}              # (W, H, device) -> [FitEngine]; an engine is checked out from forward until backward


def _checkout(W, H, n, dev):
    """An engine of the pool, reserved for the caller: returns (engine, token).  The token is the reservation --
    ``_release`` frees the engine only for the token it was checked out with, so a stale finalizer of an earlier
    forward (its graph node is collected AFTER the next forward took the same engine) cannot free it under the
    forward / backward pair that owns it now."""
    from .fused import FitEngine
    key = (int(W), int(H), str(dev))
    for eng in _POOL.setdefault(key, []):
        if not eng.busy:
            break
    else:
        eng = FitEngine(W, H, max(2 * n, 65536), dev)
        eng.busy = False
        eng.owner = None
        eng.ovf_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        eng.ovf_event = None
        _POOL[key].append(eng)
    # the pair-list overflow flag of this engine's PREVIOUS call, copied to pinned memory behind that call: read it
    # here without waiting for the device (an overflow drops splat-tile pairs silently otherwise)
    _poll_overflow(eng)
    eng.ensure_capacity(n)
    if getattr(eng, "pad2", None) is None or eng.pad2.shape[0] < eng.cap:
        eng.pad2 = torch.zeros(eng.cap, 2, dtype=torch.float32, device=eng.dev)
    eng.busy = True
    eng.owner = object()
    return eng, eng.owner


def _release(eng, token):
    if eng.owner is token:
        eng.owner = None
        eng.busy = False


def _poll_overflow(eng):
    """raises if the copy of the flag that followed the engine's last forward has landed and shows dropped pairs"""
    if eng.ovf_event is not None and eng.ovf_event.query():
        eng.ovf_event = None
        if int(eng.ovf_host[0]):
            eng.overflow.zero_()
            raise RuntimeError(f"gflow_amd.render: a render produced more than K_cap={eng.K_cap} splat-tile "
                               f"pairs and dropped some; raise K_cap (FitEngine(..., K_cap=...))")


def check_overflow():
    """Blocking: raises if ANY render of the fused operator so far dropped splat-tile pairs (the operator itself looks at
    the flag without stopping the host: at the backward of the same render, and at the next render on the same engine).
    Call it after the last render of a program whose result matters."""
    for engines in _POOL.values():
        for eng in engines:
            eng.ovf_event = None
            eng.check_overflow()


def _watch_overflow(eng):
    eng.ovf_host.copy_(eng.overflow[0:1], non_blocking=True)
    eng.ovf_event = torch.cuda.Event()
    eng.ovf_event.record()


class _FusedRender(torch.autograd.Function):
    """render(gaussians, camera) -> (rgb, depth_map, uv, depth): gfl_render_fwd / gfl_render_bwd of
    include/gflow_hip.h -- projection, covariance, EWA, binning, sort and the 4-channel composite in the fused
    kernels, one library call per direction (render.py:6-108 makes six operator calls).  The library writes straight
    into the tensors that are returned (fresh render / record buffers per call, the caller's intr / extr are read in
    place): per direction the host issues one concatenation, two allocations and the call."""

    @staticmethod
    def forward(ctx, xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H):
        dev = xyz.device
        n = xyz.shape[0]
        eng, token = _checkout(W, H, n, dev)
        try:
            if n:
                f = lambda t, c: t.detach().float().reshape(n, c)
                torch.cat([f(xyz, 3), f(scale, 3), f(rotate, 4), f(opacity, 1), f(rgb, 3), eng.pad2[:n]], dim=1,
                          out=eng.params[:n])
            intr_c = intr.detach().float().contiguous().reshape(4)
            extr_c = extr.detach().float().contiguous().reshape(12)
            out = torch.empty(4, eng.H, eng.W, dtype=torch.float32, device=dev)
            rec = torch.empty(max(n, 1), 12, dtype=torch.float32, device=dev)
            st = eng.state()
            st.N, st.intr, st.extr, st.render, st.rec = n, intr_c.data_ptr(), extr_c.data_ptr(), out.data_ptr(), rec.data_ptr()
            eng.hp.bg = float(bg)
            L.check(eng.lib.gfl_render_fwd(ctypes.byref(st), ctypes.byref(eng.hp), L.stream()), "render")
            _watch_overflow(eng)
        except Exception:
            _release(eng, token)
            raise
        if any(ctx.needs_input_grad):
            ctx.eng, ctx.n, ctx.keep, ctx.token = eng, n, (intr_c, extr_c, out, rec), token
            # a graph that is dropped without backward frees the engine too; backward() calls this same finalizer,
            # which runs at most once
            ctx.fin = weakref.finalize(ctx, _release, eng, token)
        else:
            _release(eng, token)
        return out[:3], out[3:4], rec[:n, 0:2], rec[:n, 9:10]

    @staticmethod
    def backward(ctx, d_rgb, d_depth_map, d_uv, d_depth):
        eng, n = ctx.eng, ctx.n
        if eng.owner is not ctx.token:
            # (the tile lists, records and checkpoints of the forward live in the pooled engine, which went back to the
            # pool with the first backward)
            raise RuntimeError("gflow_amd.render: backward through the fused render operator a second time "
                               "(retain_graph) is not supported; call render() again")
        _poll_overflow(eng)          # (the forward of this very render: by now its flag has usually reached the host)
        dev = eng.dev
        H, W = eng.H, eng.W
        z = lambda c: torch.zeros(c, H, W, dtype=torch.float32, device=dev)
        d_render = torch.cat([z(3) if d_rgb is None else d_rgb.float(), z(1) if d_depth_map is None else d_depth_map.float()])
        d_uv = None if d_uv is None else d_uv.float().contiguous()
        d_depth = None if d_depth is None else d_depth.float().contiguous()
        d_params = torch.empty(max(n, 1), 16, dtype=torch.float32, device=dev)
        d_extr = torch.empty(12, dtype=torch.float32, device=dev)
        L.check(eng.lib.gfl_render_bwd(ctypes.byref(eng.state()), ctypes.byref(eng.hp), L.ptr(d_render), L.ptr(d_uv),
                                       L.ptr(d_depth), L.ptr(d_params), L.ptr(d_extr), L.stream()), "render backward")
        ctx.fin()
        g = d_params[:n]
        return (g[:, 0:3], g[:, 3:6], g[:, 6:10], g[:, 10:11], g[:, 11:14], None, d_extr.reshape(3, 4), None, None, None)


def render(gaussians, camera, bg=0.0):
    """The rasteriser as ONE differentiable operator.  gaussians: dict with the ACTIVATED attributes xyz (N,3),
    scale (N,3), rotate (N,4, unit, wxyz), opacity (N,1), rgb (N,3); camera: dict with intr (4,), extr (3,4),
    W, H.  Returns dict(rgb (3,H,W), depth_map (1,H,W), uv (N,2), depth (N,1)) -- what
    render_multiple(input_group, ["rgb", "uv", "depth", "depth_map"]) returns (render.py:6-108), with gradients to
    the five attributes and to extr."""
    L.need_device(gaussians["xyz"])
    rgb, depth_map, uv, depth = _FusedRender.apply(
        gaussians["xyz"], gaussians["scale"], gaussians["rotate"], gaussians["opacity"], gaussians["rgb"],
        camera["intr"], camera["extr"], float(bg), int(camera["W"]), int(camera["H"]))
    return {"rgb": rgb, "depth_map": depth_map, "uv": uv, "depth": depth}


def render_multiple(input_group,
                    return_type=("rgb", "uv", "depth", "depth_map", "depth_map_color", "center"),
                    center_scale=10.0):
    """input_group = [xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H]
    (render.py:9).  Returns a dict with the requested entries:
    rgb (3,H,W), uv (N,2), depth (N,1), depth_map (1,H,W), depth_map_color (3,H,W),
    center (3,H,W)."""
    xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H = input_group
    if USE_FUSED and xyz.is_cuda and set(return_type) <= FUSED_TYPES and ("rgb" in return_type or "depth_map" in return_type):
        # the training call (trainer.py:404-407 minus the two snapshot images): one fused operator
        full = render(dict(xyz=xyz, scale=scale, rotate=rotate, opacity=opacity, rgb=rgb),
                      dict(intr=intr, extr=extr, W=W, H=H), bg)
        return {k: full[k] for k in return_type}
    out = {}
    uv, depth = msplat.project_point(xyz, intr, extr, W, H)
    visible = depth != 0
    if "uv" in return_type:
        out["uv"] = uv
    if "depth" in return_type:
        out["depth"] = depth
    cov3d = msplat.compute_cov3d(scale, rotate, visible)
    conic, radius, tiles_touched = msplat.ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
    ids, tile_range = msplat.sort_gaussian(uv, depth, W, H, radius, tiles_touched)
    if "rgb" in return_type:
        out["rgb"] = msplat.alpha_blending(uv, conic, opacity, rgb, ids, tile_range, bg, W, H)
    if "depth_map" in return_type:
        out["depth_map"] = msplat.alpha_blending(uv, conic, opacity, depth, ids, tile_range, bg, W, H)
    if "depth_map_color" in return_type:
        depth_color = apply_float_colormap(depth, colormap="turbo", non_zero=True)
        out["depth_map_color"] = msplat.alpha_blending(uv, conic, opacity, depth_color, ids, tile_range, bg, W, H)
    if "center" in return_type:
        # unit-variance blobs at the splat centres, same sorted lists (render.py:93-106)
        unit = torch.tensor([1.0, 0.0, 1.0], device=conic.device)
        out["center"] = msplat.alpha_blending(uv, torch.ones_like(conic) * unit, torch.ones_like(opacity), rgb, ids,
                                              tile_range, bg, W, H)
    return out


def render_traj(input_group, point_num, line_scale=1.0, point_scale=2.0):
    """Trajectory overlay (render.py:110-156): isotropic blobs, the last
    ``point_num`` splats drawn with ``line_scale``, the others with ``point_scale``."""
    xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H = input_group
    uv, depth = msplat.project_point(xyz, intr, extr, W, H)
    visible = depth != 0
    cov3d = msplat.compute_cov3d(scale, rotate, visible)
    conic, radius, tiles_touched = msplat.ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
    ids, tile_range = msplat.sort_gaussian(uv, depth, W, H, radius, tiles_touched)
    unit = torch.tensor([1.0, 0.0, 1.0], device=conic.device)
    conic = torch.ones_like(conic) * unit * line_scale
    conic[:-point_num] = torch.ones_like(conic[:-point_num]) * unit * point_scale
    return msplat.alpha_blending(uv, conic, opacity, rgb, ids, tile_range, bg, W, H)


def render2img_device(rendered):
    """(3,H,W) float -> (H,W,3) uint8 tensor ON THE DEVICE: the clamp / x255 / truncation of render.py:158-166
    without the host round trip (a snapshot used to cost ~6 ms of numpy arithmetic on three 1.2 M-element
    float images; the iteration it interrupts takes 0.22 ms).  Nothing here waits for the GPU."""
    return (torch.clamp(rendered.detach(), 0.0, 1.0) * 255.0).to(torch.uint8).permute(1, 2, 0).contiguous()


def render2img(rendered):
    """(3,H,W) float -> (H,W,3) uint8 numpy (render.py:158-166)."""
    return render2img_device(rendered).cpu().numpy()
