"""Small seeded splat scenes shared by the oracle and GPU parity tests."""
import math

import numpy as np
import torch


def camera(W, H, f=None, dtype=torch.float32, tilt=False):
    f = float(f if f is not None else 0.6 * W)
    intr = torch.tensor([f, f, W / 2.0, H / 2.0], dtype=dtype)
    if tilt:
        a, b = 0.07, -0.05
        Ry = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=dtype)
        Rx = torch.tensor([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]], dtype=dtype)
        extr = torch.cat([Ry @ Rx, torch.tensor([[0.05], [-0.03], [0.1]], dtype=dtype)], dim=1)
    else:
        extr = torch.eye(4, dtype=dtype)[:3].clone()
    return intr, extr


def random_scene(N, W, H, seed=0, dtype=torch.float32, sigma_px=2.0, spread=1.15, tilt=True, behind=0.05):
    """N splats scattered over (and a little beyond) the frustum, depth 1..4,
    projected sigma about ``sigma_px`` pixels (log-normal), random rotations,
    opacities in (0.05,0.999).  A fraction ``behind`` sits behind the camera."""
    g = torch.Generator().manual_seed(seed)
    intr, extr = camera(W, H, dtype=dtype, tilt=tilt)
    f = intr[0].item()
    z = 1.0 + 3.0 * torch.rand(N, generator=g, dtype=torch.float64)
    u = (torch.rand(N, generator=g, dtype=torch.float64) - 0.5) * W * spread + W / 2
    v = (torch.rand(N, generator=g, dtype=torch.float64) - 0.5) * H * spread + H / 2
    x = (u - W / 2) / f * z
    y = (v - H / 2) / f * z
    nb = int(N * behind)
    if nb:
        z[:nb] = -z[:nb]
    xyz_cam = torch.stack([x, y, z], dim=1)
    R = extr[:, :3].double()
    t = extr[:, 3].double()
    xyz = (xyz_cam - t) @ R                       # inverse of R x + t  (R orthonormal)
    sig = sigma_px * torch.exp(0.5 * torch.randn(N, generator=g, dtype=torch.float64))
    aniso = torch.exp(0.4 * torch.randn(N, 3, generator=g, dtype=torch.float64))
    scale = (sig / f * z.abs()).unsqueeze(1) * aniso
    rot = torch.nn.functional.normalize(torch.randn(N, 4, generator=g, dtype=torch.float64), dim=1)
    opacity = 0.05 + 0.949 * torch.rand(N, 1, generator=g, dtype=torch.float64)
    rgb = torch.rand(N, 3, generator=g, dtype=torch.float64)
    cast = lambda a: a.to(dtype).contiguous()
    return dict(xyz=cast(xyz), scale=cast(scale), rotate=cast(rot), opacity=cast(opacity), rgb=cast(rgb),
                intr=intr, extr=extr, W=W, H=H)


def scene_group(s, bg=0.0):
    return [s["xyz"], s["scale"], s["rotate"], s["opacity"], s["rgb"], s["intr"], s["extr"], bg, s["W"], s["H"]]
