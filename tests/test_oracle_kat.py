"""Known-answer and self-consistency tests of the rasteriser oracle (CPU).
The rasteriser oracle is PARITY-UNPINNED against real msplat (oracle/__init__.py);
these tests pin it analytically instead (last row)."""
import math

import numpy as np
import pytest
import torch

from oracle import msplat_oracle as MO
from tests.scenes import camera, random_scene, scene_group


def _one_blob(W, H, u, v, conic, opacity, feat, bg, C):
    uv = torch.tensor([[u, v]], dtype=torch.float32)
    con = torch.tensor([conic], dtype=torch.float32)
    op = torch.tensor([[opacity]], dtype=torch.float32)
    f = torch.tensor([feat], dtype=torch.float32)
    ids, tr = MO.sort_gaussian(uv, torch.tensor([[1.0]]), W, H, torch.tensor([[6]], dtype=torch.int32),
                               torch.tensor([[0]], dtype=torch.int32))
    return MO.alpha_blending(uv, con, op, f, ids, tr, bg, W, H)


def test_center_recipe_single_blob():
    # the reference's own "center" recipe: conic [1,0,1], opacity 1 (render.py:93-97)
    W, H = 48, 32
    out = _one_blob(W, H, 20.0, 12.0, [1.0, 0.0, 1.0], 1.0, [0.2, 0.5, 0.9], 0.33, 3)
    tile_x0, tile_x1 = 0, 32    # radius 6 around u=20 touches tile columns 0 and 1
    for (x, y) in [(20, 12), (21, 12), (22, 14), (17, 9), (26, 12)]:
        r2 = (20 - x) ** 2 + (12 - y) ** 2
        a = min(0.99, math.exp(-0.5 * r2))
        a = a if a >= 1 / 255 else 0.0
        for c, f in enumerate([0.2, 0.5, 0.9]):
            assert abs(out[c, y, x].item() - (a * f + (1 - a) * 0.33)) < 1e-6
    # far away: background only
    assert abs(out[0, 30, 46].item() - 0.33) < 1e-7


def test_depth_order_matters():
    W, H = 32, 32
    uv = torch.tensor([[10.0, 10.0], [11.0, 10.0]])
    con = torch.tensor([[0.3, 0.0, 0.3], [0.3, 0.0, 0.3]])
    op = torch.tensor([[0.8], [0.8]])
    f = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    rad = torch.tensor([[8], [8]], dtype=torch.int32)
    tl = torch.zeros(2, 1, dtype=torch.int32)
    imgs = []
    for d in ([[1.0], [2.0]], [[2.0], [1.0]]):
        ids, tr = MO.sort_gaussian(uv, torch.tensor(d), W, H, rad, tl)
        imgs.append(MO.alpha_blending(uv, con, op, f, ids, tr, 0.0, W, H))
    a0 = 0.8
    a1 = 0.8 * math.exp(-0.5 * 0.3)
    # pixel (10,10): splat 0 centred there
    assert abs(imgs[0][0, 10, 10].item() - a0) < 1e-6
    assert abs(imgs[0][1, 10, 10].item() - a1 * (1 - a0)) < 1e-6
    assert abs(imgs[1][1, 10, 10].item() - a1) < 1e-6
    assert abs(imgs[1][0, 10, 10].item() - a0 * (1 - a1)) < 1e-6


def test_culled_points_signal_zero():
    W, H = 64, 48
    intr, extr = camera(W, H)
    xyz = torch.tensor([[0.0, 0.0, -1.0], [0.0, 0.0, 0.1], [0.0, 0.0, 2.0], [100.0, 0.0, 1.0]])
    uv, depth = MO.project_point(xyz, intr, extr, W, H)
    assert depth[0, 0] == 0 and depth[1, 0] == 0 and depth[3, 0] == 0
    assert torch.all(uv[0] == 0) and torch.all(uv[1] == 0) and torch.all(uv[3] == 0)
    assert depth[2, 0] == 2.0 and abs(uv[2, 0].item() - W / 2) < 1e-5 and abs(uv[2, 1].item() - H / 2) < 1e-5


def test_depth_channel_with_background():
    W, H = 32, 16
    out = _one_blob(W, H, 8.0, 8.0, [0.5, 0.0, 0.5], 0.6, [2.5], 1.0, 1)
    a = 0.6
    assert abs(out[0, 8, 8].item() - (a * 2.5 + (1 - a) * 1.0)) < 1e-6
    assert out.shape == (1, H, W)


def test_empty_input_gives_background():
    W, H = 40, 24
    s = random_scene(0, W, H)
    out = MO.render_multiple(scene_group(s, bg=0.33), ["rgb", "depth_map", "uv"])
    assert out["rgb"].shape == (3, H, W) and torch.all(out["rgb"] == 0.33)
    assert torch.all(out["depth_map"] == 0.33) and out["uv"].shape == (0, 2)


def test_identity_cov_projects_to_isotropic_conic():
    # unit-quaternion, isotropic scale s at depth z on the axis: Sigma2 = (f s / z)^2 I + 0.3 I
    W, H = 64, 64
    intr, extr = camera(W, H, f=50.0)
    xyz = torch.tensor([[0.0, 0.0, 2.0]])
    sc = torch.tensor([[0.1, 0.1, 0.1]])
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]])
    uv, depth = MO.project_point(xyz, intr, extr, W, H)
    cov = MO.compute_cov3d(sc, q, depth != 0)
    np.testing.assert_allclose(cov.numpy(), [[0.01, 0, 0, 0.01, 0, 0.01]], atol=1e-8)
    conic, radius, tiles = MO.ewa_project(xyz, cov, intr, extr, uv, W, H, depth != 0)
    var = (50.0 * 0.1 / 2.0) ** 2 + 0.3
    np.testing.assert_allclose(conic.numpy(), [[1 / var, 0, 1 / var]], rtol=1e-5, atol=1e-7)
    assert radius.item() == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    assert tiles.item() == 4          # centre (32,32) radius 8 -> tiles 1..2 in x and y


@pytest.mark.parametrize("seed", [0, 1])
def test_vectorised_blend_matches_literal_loops(seed):
    W, H = 40, 36                      # ragged: 2.5 x 2.25 tiles
    s = random_scene(300, W, H, seed=seed, sigma_px=3.0)
    uv, depth = MO.project_point(s["xyz"], s["intr"], s["extr"], W, H)
    vis = depth != 0
    cov = MO.compute_cov3d(s["scale"], s["rotate"], vis)
    conic, radius, tiles = MO.ewa_project(s["xyz"], cov, s["intr"], s["extr"], uv, W, H, vis)
    ids, tr = MO.sort_gaussian(uv, depth, W, H, radius, tiles)
    assert ids.numel() == int(tiles.sum())
    feat = torch.cat([s["rgb"], depth], dim=1)
    fast = MO.alpha_blending(uv, conic, s["opacity"], feat, ids, tr, 0.33, W, H)
    slow, final_T, ncon = MO.alpha_blending_loops(uv, conic, s["opacity"], feat, ids, tr, 0.33, W, H)
    np.testing.assert_allclose(fast.numpy(), slow, rtol=2e-5, atol=2e-6)
    assert ncon.max() > 3             # the scene really has overlapping splats


def test_termination_branch_is_exercised():
    # a stack of opaque splats: T falls below 1e-4 and later splats must not contribute
    W, H = 16, 16
    n = 12
    uv = torch.full((n, 2), 8.0)
    con = torch.tensor([[0.05, 0.0, 0.05]]).repeat(n, 1)
    op = torch.full((n, 1), 0.95)
    f = torch.linspace(0.1, 1.0, n).unsqueeze(1)
    d = torch.linspace(1.0, 2.0, n).unsqueeze(1)
    ids, tr = MO.sort_gaussian(uv, d, W, H, torch.full((n, 1), 10, dtype=torch.int32), torch.zeros(n, 1, dtype=torch.int32))
    fast = MO.alpha_blending(uv, con, op, f, ids, tr, 1.0, W, H)
    slow, final_T, ncon = MO.alpha_blending_loops(uv, con, op, f, ids, tr, 1.0, W, H)
    np.testing.assert_allclose(fast.numpy(), slow, rtol=1e-5, atol=1e-7)
    assert ncon[8, 8] == 3             # 0.05^3 = 1.25e-4 >= 1e-4 > 0.05^4
    assert abs(final_T[8, 8] - 0.05 ** 3) < 1e-9


def test_gradcheck_float64_ops():
    W, H = 32, 32
    s = random_scene(6, W, H, seed=3, dtype=torch.float64, sigma_px=3.0, behind=0.0, spread=0.6)
    xyz = s["xyz"].clone().requires_grad_(True)
    extr = s["extr"].clone().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, e: MO.project_point(a, s["intr"], e, W, H), (xyz, extr), atol=1e-7)
    sc = s["scale"].clone().requires_grad_(True)
    q = s["rotate"].clone().requires_grad_(True)
    vis = torch.ones(6, 1, dtype=torch.bool)
    assert torch.autograd.gradcheck(lambda a, b: MO.compute_cov3d(a, b, vis), (sc, q), atol=1e-7)
    uv, depth = MO.project_point(s["xyz"], s["intr"], s["extr"], W, H)
    cov = MO.compute_cov3d(s["scale"], s["rotate"], vis).clone().requires_grad_(True)
    assert torch.autograd.gradcheck(
        lambda a, c, e: MO.ewa_project(a, c, s["intr"], e, uv, W, H, vis)[0], (xyz, cov, extr), atol=1e-6)


def test_gradcheck_float64_full_render():
    W, H = 32, 32
    s = random_scene(5, W, H, seed=5, dtype=torch.float64, sigma_px=3.0, behind=0.0, spread=0.5)
    leaves = [s[k].clone().requires_grad_(True) for k in ("xyz", "scale", "rotate", "opacity", "rgb")]
    extr = s["extr"].clone().requires_grad_(True)
    wts = torch.rand(4, H, W, dtype=torch.float64, generator=torch.Generator().manual_seed(0))

    def fn(xyz, scale, rot, op, rgb, e):
        o = MO.render_multiple([xyz, scale, rot, op, rgb, s["intr"], e, 0.2, W, H], ["rgb", "depth_map"])
        return (torch.cat([o["rgb"], o["depth_map"]]) * wts).sum()

    assert torch.autograd.gradcheck(fn, (*leaves, extr), atol=1e-6, rtol=1e-4, nondet_tol=0.0)


def test_sh_basis_is_orthonormal_on_the_sphere():
    """The only pin available for A17 (no reference call site): the 16 real harmonics are
    orthonormal, integral over the unit sphere of Y_i Y_j = delta_ij (Gauss-Legendre x uniform
    azimuth quadrature, exact for these polynomial degrees), and degree 0 is the constant
    1 / (2 sqrt(pi))."""
    nz, nphi = 16, 32
    zs, wz = np.polynomial.legendre.leggauss(nz)
    phi = (np.arange(nphi) + 0.5) * 2 * np.pi / nphi
    Z, PHI = np.meshgrid(zs, phi, indexing="ij")
    R = np.sqrt(1 - Z ** 2)
    dirs = torch.tensor(np.stack([R * np.cos(PHI), R * np.sin(PHI), Z], -1).reshape(-1, 3))
    w = torch.tensor(np.repeat(wz, nphi) * 2 * np.pi / nphi)
    B = MO.sh_basis(dirs, 16)                                    # (Q,16) float64
    gram = (B * w.unsqueeze(1)).T @ B
    np.testing.assert_allclose(gram.numpy(), np.eye(16), atol=1e-12)
    assert abs(B[0, 0].item() - 0.5 / np.sqrt(np.pi)) < 1e-15


def test_sh_degree_zero_is_constant_and_visible_masks():
    g = torch.Generator().manual_seed(1)
    shs = torch.randn(5, 4, 3, generator=g, dtype=torch.float64)
    d = torch.nn.functional.normalize(torch.randn(5, 3, generator=g, dtype=torch.float64), dim=1)
    out0 = MO.compute_sh(shs[:, :1], d)
    np.testing.assert_allclose(out0.numpy(), (MO.SH_C0 * shs[:, 0]).numpy(), rtol=1e-15)
    vis = torch.tensor([1, 0, 1, 1, 0], dtype=torch.bool).reshape(5, 1)
    out = MO.compute_sh(shs, d, vis)
    assert torch.all(out[~vis.reshape(-1)] == 0)
