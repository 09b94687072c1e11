"""Capture what REAL msplat computes on this repository's known-answer scenes -> tests/golden/msplat_ref.npz.

Why: the rasteriser half of the oracle (oracle/msplat_oracle.py) is PARITY-UNPINNED -- msplat is a CUDA extension that is
not vendored in GFlow and cannot be built or imported where this repository is developed.  Every constant
that is internal to msplat is an assumption (include/gflow_hip.h GFL_*).  Somebody WITH msplat closes that in one command:

    # on a CUDA box where `import msplat` works (pip install of github.com/pointrix-project/msplat), from the repo root:
    python tools/capture_msplat_golden.py                    # writes tests/golden/msplat_ref.npz
    python -m pytest tests/test_oracle_golden.py -k msplat   # the oracle against it (CPU)
    # copy the .npz to the MI355X box:  python -m pytest tests/test_gpu_msplat_golden.py -m gpu   (the HIP path against it)

The five operators are called exactly as /root/reference/gflow/utils/render.py does (positional arguments, in this order):
    :21-24  uv, depth = msplat.project_point(xyz, intr, extr, W, H)        :29  visible = depth != 0
    :37-41  cov3d = msplat.compute_cov3d(scale, rotate, visible)
    :44-49  conic, radius, tiles_touched = msplat.ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
    :52-54  gaussian_ids_sorted, tile_range = msplat.sort_gaussian(uv, depth, W, H, radius, tiles_touched)
    :58-64  rendered = msplat.alpha_blending(uv, conic, opacity, feature, gaussian_ids_sorted, tile_range, bg, W, H)
forward and backward (one scalar loss with seeded weights over rgb, depth_map, uv and depth: every gradient the reference
consumes -- xyz, scale, rotate, opacity, rgb, extr), plus the "center" composite of render.py:93-106.

``--module gflow_amd.msplat`` runs the same script against THIS repository's operators: a FORMAT CHECK of the script and of
the two tests (the file then says so in ``meta_module`` and the tests refuse to count it as a pin -- they only check that the
arrays are all there); it needs a HIP device.  The oracle itself can be captured with ``--module oracle.msplat_oracle`` (CPU):
that is how the tests' plumbing is exercised where neither msplat nor a GPU exists.

The scenes are data made by tests/scenes.py from seeds (committed); nothing of the reference is copied.
"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.scenes import camera, random_scene  # noqa: E402

NAMES = ("xyz", "scale", "rotate", "opacity", "rgb")
OUT_DEFAULT = os.path.join(ROOT, "tests", "golden", "msplat_ref.npz")


def scenes():
    """name -> (scene dict, bg).  The known-answer scenes of tests/test_oracle_kat.py as 3-D splats (so that all five
    operators run on them) and the random scene the fused-operator parity test uses."""
    out = {}
    # one isotropic splat on the optical axis (test_identity_cov_projects_to_isotropic_conic)
    W, H = 64, 64
    intr, extr = camera(W, H, f=50.0)
    out["one_blob"] = (dict(xyz=torch.tensor([[0.0, 0.0, 2.0]]), scale=torch.tensor([[0.1, 0.1, 0.1]]),
                            rotate=torch.tensor([[1.0, 0.0, 0.0, 0.0]]), opacity=torch.tensor([[0.9]]),
                            rgb=torch.tensor([[0.2, 0.5, 0.9]]), intr=intr, extr=extr, W=W, H=H), 0.33)
    # one splat at a NON-integer pixel position, narrow: the pixel-centre convention (GFL_PIXEL_CENTER) shows in which pixel is brightest
    W, H = 48, 32
    intr, extr = camera(W, H, f=40.0)
    u, v, z = 20.25, 12.75, 2.0
    out["off_centre_blob"] = (dict(xyz=torch.tensor([[(u - W / 2) / 40.0 * z, (v - H / 2) / 40.0 * z, z]]),
                                   scale=torch.tensor([[0.04, 0.04, 0.04]]), rotate=torch.tensor([[1.0, 0.0, 0.0, 0.0]]),
                                   opacity=torch.tensor([[0.95]]), rgb=torch.tensor([[1.0, 0.5, 0.25]]),
                                   intr=intr, extr=extr, W=W, H=H), 0.0)
    # two overlapping splats, depth ordered (test_depth_order_matters), and four culled / edge points
    # (test_culled_points_signal_zero: behind the camera, inside the near plane, far outside the frustum)
    W, H = 64, 48
    intr, extr = camera(W, H)
    f = float(intr[0])
    pts = [[(10 - W / 2) / f * 1.0, (10 - H / 2) / f * 1.0, 1.0], [(11 - W / 2) / f * 2.0, (10 - H / 2) / f * 2.0, 2.0],
           [0.0, 0.0, -1.0], [0.0, 0.0, 0.1], [100.0, 0.0, 1.0], [0.0, 0.0, 0.21]]
    n = len(pts)
    out["order_and_culling"] = (dict(xyz=torch.tensor(pts), scale=torch.full((n, 3), 0.05),
                                     rotate=torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(n, 1),
                                     opacity=torch.full((n, 1), 0.8),
                                     rgb=torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [1, 1, 0], [0, 1, 1], [1, 0, 1.0]]),
                                     intr=intr, extr=extr, W=W, H=H), 0.0)
    # opaque stack: the T < 1e-4 stop rule and the 0.99 cap (test_termination_branch_is_exercised)
    W, H = 32, 32
    intr, extr = camera(W, H, f=30.0)
    n = 12
    z = torch.linspace(1.0, 3.0, n)
    out["opaque_stack"] = (dict(xyz=torch.stack([torch.zeros(n), torch.zeros(n), z], 1),
                                scale=(0.2 * z).unsqueeze(1).repeat(1, 3), rotate=torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(n, 1),
                                opacity=torch.full((n, 1), 0.999), rgb=torch.rand(n, 3, generator=torch.Generator().manual_seed(2)),
                                intr=intr, extr=extr, W=W, H=H), 1.0)
    # the scene of tests/test_gpu_render_op.py (3 000 splats, tilted camera, a share behind the camera), ragged tile grid
    out["random_3000"] = (random_scene(3000, 200, 136, seed=11, sigma_px=2.5), 0.33)
    return out


def run_scene(ms, s, bg, dev):
    """The five calls of render.py:21-64 (+ the center composite, :93-106) forward and backward; everything as numpy."""
    W, H = int(s["W"]), int(s["H"])
    leaves = {k: s[k].clone().float().to(dev).requires_grad_(True) for k in NAMES}
    intr = s["intr"].float().to(dev)
    extr = s["extr"].clone().float().to(dev).requires_grad_(True)
    xyz, scale, rotate, opacity, rgb = (leaves[k] for k in NAMES)
    uv, depth = ms.project_point(xyz, intr, extr, W, H)
    visible = depth != 0
    cov3d = ms.compute_cov3d(scale, rotate, visible)
    conic, radius, tiles_touched = ms.ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
    ids, tile_range = ms.sort_gaussian(uv, depth, W, H, radius, tiles_touched)
    img = ms.alpha_blending(uv, conic, opacity, rgb, ids, tile_range, bg, W, H)
    depth_map = ms.alpha_blending(uv, conic, opacity, depth, ids, tile_range, bg, W, H)
    with torch.no_grad():
        unit = torch.tensor([1.0, 0.0, 1.0], device=conic.device)
        center = ms.alpha_blending(uv, torch.ones_like(conic) * unit, torch.ones_like(opacity), rgb, ids, tile_range, bg, W, H)
    n = xyz.shape[0]
    g = torch.Generator().manual_seed(5)
    w_rgb, w_dm = torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g)
    w_uv, w_d = 1e-2 * torch.randn(n, 2, generator=g), 1e-1 * torch.randn(n, 1, generator=g)
    loss = ((img * w_rgb.to(dev)).sum() + (depth_map * w_dm.to(dev)).sum() + (uv * w_uv.to(dev)).sum()
            + (depth * w_d.to(dev)).sum())
    loss.backward()
    np_ = lambda t: t.detach().cpu().numpy()
    out = {"in_" + k: np_(s[k].float()) for k in NAMES}
    out.update(in_intr=np_(s["intr"].float()), in_extr=np_(s["extr"].float()), in_bg=np.float32(bg),
               in_W=np.int32(W), in_H=np.int32(H), w_rgb=np_(w_rgb), w_depth_map=np_(w_dm), w_uv=np_(w_uv), w_depth=np_(w_d),
               uv=np_(uv), depth=np_(depth), cov3d=np_(cov3d), conic=np_(conic),
               radius=np_(radius).astype(np.int32), tiles_touched=np_(tiles_touched).astype(np.int32),
               ids=np_(ids).astype(np.int32), tile_range=np_(tile_range).astype(np.int32).reshape(-1, 2),
               rgb=np_(img), depth_map=np_(depth_map), center=np_(center), loss=np.float64(loss.item()),
               d_extr=np_(extr.grad))
    for k in NAMES:
        out["d_" + k] = np_(leaves[k].grad)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--module", default="msplat", help="the module that provides the five operators (default: real msplat)")
    ap.add_argument("--out", default=OUT_DEFAULT)
    ap.add_argument("--device", default=None, help="default: cuda if available (msplat is CUDA only), cpu for the oracle")
    args = ap.parse_args(argv)
    ms = importlib.import_module(args.module)
    dev = args.device or ("cpu" if args.module.startswith("oracle") else "cuda")
    blob = {"meta_module": np.array(args.module), "meta_module_version": np.array(str(getattr(ms, "__version__", "?"))),
            "meta_torch": np.array(torch.__version__), "meta_device": np.array(
                torch.cuda.get_device_name(0) if (dev != "cpu" and torch.cuda.is_available()) else "cpu"),
            # only a capture of the real extension pins anything
            "meta_is_reference": np.array(args.module == "msplat")}
    names = []
    for name, (s, bg) in scenes().items():
        for k, v in run_scene(ms, s, bg, dev).items():
            blob[f"{name}__{k}"] = v
        names.append(name)
        print(f"{name}: {s['xyz'].shape[0]} splats, {int(s['W'])}x{int(s['H'])}, K = {blob[name + '__ids'].shape[0]} pairs")
    blob["meta_scenes"] = np.array(names)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **blob)
    print(f"wrote {args.out} ({os.path.getsize(args.out) / 1e6:.2f} MB) from module {args.module!r}"
          + ("" if args.module == "msplat" else "  -- NOT the reference: a format check, it pins nothing"))


if __name__ == "__main__":
    main()
