"""Render orchestrator -- the counterpart of gflow/utils/render.py.

``render_multiple`` / ``render_traj`` / ``render2img`` keep the reference's
signatures and semantics (render.py:6-166) on top of ``gflow_amd.msplat``.
"""
import numpy as np
import torch

from . import msplat
from .color import apply_float_colormap


def render_multiple(input_group,
                    return_type=("rgb", "uv", "depth", "depth_map", "depth_map_color", "center"),
                    center_scale=10.0):
    """input_group = [xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H]
    (render.py:9).  Returns a dict with the requested entries:
    rgb (3,H,W), uv (N,2), depth (N,1), depth_map (1,H,W), depth_map_color (3,H,W),
    center (3,H,W)."""
    xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H = input_group
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


def render2img(rendered):
    """(3,H,W) float -> (H,W,3) uint8 numpy (render.py:158-166)."""
    rendered = torch.clamp(rendered.detach().permute(1, 2, 0), 0.0, 1.0)
    return (rendered.cpu().numpy() * 255).astype(np.uint8)
