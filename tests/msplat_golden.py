"""What tests/test_oracle_golden.py (the oracle, CPU) and tests/test_gpu_msplat_golden.py (the HIP path) share: reading a
capture written by tools/capture_msplat_golden.py and holding an implementation of the five operators against it.
North_star's bar: rendered values within 1e-4 relative of the reference rasteriser on identical splats and camera."""
import os

import numpy as np
import torch

from tests.test_gpu_parity import close_frac

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PATH = os.path.join(ROOT, "tests", "golden", "msplat_ref.npz")
NAMES = ("xyz", "scale", "rotate", "opacity", "rgb")
KEYS = ("uv", "depth", "cov3d", "conic", "radius", "tiles_touched", "ids", "tile_range", "rgb", "depth_map", "center", "loss",
        "d_extr") + tuple("d_" + k for k in NAMES)


def load(path=REF_PATH):
    g = np.load(path)
    meta = {k[5:]: (g[k].item() if g[k].shape == () else g[k].tolist()) for k in g.files if k.startswith("meta_")}
    return g, meta


def check_complete(g, meta):
    for name in meta["scenes"]:
        for k in KEYS + tuple("in_" + k for k in NAMES) + ("in_intr", "in_extr", "in_bg", "in_W", "in_H", "w_rgb", "w_depth_map",
                                                          "w_uv", "w_depth"):
            assert f"{name}__{k}" in g.files, f"{name}__{k} missing from the capture"


def hold(ms, g, meta, dev, fused_render=None):
    """Run the five operators of module ``ms`` on every captured scene (as render.py:21-64 calls them) and compare with the
    capture.  ``fused_render(leaves, intr, extr, bg, W, H) -> dict(rgb, depth_map, uv, depth)``: also hold the fused operator
    (values and gradients).  Returns the per-scene report lines."""
    report = []
    for name in meta["scenes"]:
        r = lambda k: g[f"{name}__{k}"]
        W, H, bg = int(r("in_W")), int(r("in_H")), float(r("in_bg"))
        t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
        intr = t(r("in_intr"))
        w = [t(r(k)) for k in ("w_rgb", "w_depth_map", "w_uv", "w_depth")]

        def loss_of(o):
            return (o["rgb"] * w[0]).sum() + (o["depth_map"] * w[1]).sum() + (o["uv"] * w[2]).sum() + (o["depth"] * w[3]).sum()

        def grads_close(leaves, extr, tag):
            worst = 0.0
            for k in NAMES + ("extr",):
                ref = torch.from_numpy(r("d_" + k)).double()
                got = (extr if k == "extr" else leaves[k]).grad.detach().cpu().double()
                if ref.norm() == 0:
                    assert got.abs().max() <= 1e-6, f"{name} {tag}: d_{k} should be zero"
                    continue
                rel = float((got - ref).norm() / ref.norm())
                worst = max(worst, rel)
                assert rel < 2e-3, f"{name} {tag}: d_{k} relative L2 error {rel:.2e}"
            return worst

        leaves = {k: t(r("in_" + k)).requires_grad_(True) for k in NAMES}
        extr = t(r("in_extr")).requires_grad_(True)
        xyz, scale, rotate, opacity, rgb = (leaves[k] for k in NAMES)
        uv, depth = ms.project_point(xyz, intr, extr, W, H)
        visible = depth != 0
        cov3d = ms.compute_cov3d(scale, rotate, visible)
        conic, radius, tiles = ms.ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
        ids, tile_range = ms.sort_gaussian(uv, depth, W, H, radius, tiles)
        out = dict(uv=uv, depth=depth,
                   rgb=ms.alpha_blending(uv, conic, opacity, rgb, ids, tile_range, bg, W, H),
                   depth_map=ms.alpha_blending(uv, conic, opacity, depth, ids, tile_range, bg, W, H))
        with torch.no_grad():
            unit = torch.tensor([1.0, 0.0, 1.0], device=conic.device)
            center = ms.alpha_blending(uv, torch.ones_like(conic) * unit, torch.ones_like(opacity), rgb, ids, tile_range, bg, W, H)
        # the culling signal first: the same splats visible (render.py:29)
        vis_ref = r("depth").reshape(-1) != 0
        assert np.array_equal(visible.reshape(-1).cpu().numpy(), vis_ref), f"{name}: another set of splats is culled"
        close_frac(uv, r("uv"), 1e-4, 1e-3, what=f"{name} uv")
        close_frac(depth, r("depth"), 1e-5, 1e-6, what=f"{name} depth")
        close_frac(cov3d, r("cov3d"), 1e-4, 1e-9, what=f"{name} cov3d")
        v = torch.from_numpy(vis_ref)
        close_frac(conic.detach().cpu()[v], r("conic")[vis_ref], 1e-4, 1e-7, what=f"{name} conic")
        for key, got in (("radius", radius), ("tiles_touched", tiles)):
            a, b = got.reshape(-1).cpu().numpy().astype(np.int64), r(key).reshape(-1).astype(np.int64)
            flips = int((a != b).sum())
            assert flips <= max(1, int(1e-3 * a.size)) and (np.abs(a - b).max() if a.size else 0) <= max(1, int(np.abs(b).max() // 8)), \
                f"{name}: {key} differs at {flips} of {a.size} splats"
        for key, got in (("rgb", out["rgb"]), ("depth_map", out["depth_map"]), ("center", center)):
            close_frac(got, r(key), 1e-4, 1e-5, bad_frac=3e-4, hard=2e-2, what=f"{name} {key}")
        loss_of(out).backward()
        worst = grads_close(leaves, extr, "operators")
        line = f"{name}: five operators within 1e-4 of the capture, gradients <= {worst:.1e} rel"
        if fused_render is not None:
            leaves_f = {k: t(r("in_" + k)).requires_grad_(True) for k in NAMES}
            extr_f = t(r("in_extr")).requires_grad_(True)
            of = fused_render(leaves_f, intr, extr_f, bg, W, H)
            for key in ("rgb", "depth_map"):
                close_frac(of[key], r(key), 1e-4, 1e-5, bad_frac=3e-4, hard=2e-2, what=f"{name} fused {key}")
            close_frac(of["uv"], r("uv"), 1e-4, 1e-3, what=f"{name} fused uv")
            loss_of(of).backward()
            worst = grads_close(leaves_f, extr_f, "fused operator")
            line += f"; fused operator too (gradients <= {worst:.1e})"
        report.append(line)
    return report
