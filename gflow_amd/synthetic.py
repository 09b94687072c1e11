"""Seeded synthetic frames and clips for benchmarks and tests.

There is no DAVIS data in the build container or on the GPU box, so the workload is
generated: an image that is a sum of random-phase sinusoids plus soft discs (so the
Sobel magnitude is non-degenerate), a depth plane with smooth bumps in [1,5], a flow
field, a disc-shaped moving region and a pinhole camera with fx=fy=500."""
import math

import numpy as np
import torch


def make_frame(H=480, W=854, seed=0, shift=(0.0, 0.0)):
    """Returns dict(image (H,W,3), depth (H,W,1), flow (H,W,2), move_mask (H,W) bool,
    occ_mask (H,W) bool, focal, pp).  ``shift`` translates the texture (pixels), used
    to build clips with camera / object motion."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xs, ys = xx + shift[0], yy + shift[1]
    img = np.zeros((H, W, 3))
    for _ in range(6):
        fx, fy = rng.uniform(0.005, 0.05, 2)
        ph = rng.uniform(0, 2 * math.pi, 3)
        amp = rng.uniform(0.05, 0.2, 3)
        for c in range(3):
            img[..., c] += amp[c] * np.sin(2 * math.pi * (fx * xs + fy * ys) + ph[c])
    for _ in range(20):
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(10, 60)
        col = rng.uniform(-0.4, 0.4, 3)
        soft = 1.0 / (1.0 + np.exp((np.sqrt((xs - cx) ** 2 + (ys - cy) ** 2) - r) / 2.0))
        img += soft[..., None] * col
    img = np.clip(0.5 + img, 0.0, 1.0)
    depth = 2.0 + 0.002 * (xs - W / 2) + 0.5 * np.sin(xs / 90.0) * np.cos(ys / 70.0) + 0.3 * np.sin(ys / 40.0)
    depth = np.clip(depth, 1.0, 5.0)
    flow = np.stack([1.5 + np.sin(ys / 60.0), -0.5 + np.cos(xs / 80.0)], axis=-1)
    mcx, mcy, mr = 0.6 * W + shift[0], 0.45 * H + shift[1], math.sqrt(0.10 * H * W / math.pi)
    move = ((xx - mcx) ** 2 + (yy - mcy) ** 2) < mr ** 2
    occ = ((xx - (mcx - mr)) ** 2 / (0.15 * mr) ** 2 + (yy - mcy) ** 2 / mr ** 2) < 1.0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    return dict(image=t(img), depth=t(depth).unsqueeze(-1), flow=t(flow), move_mask=torch.from_numpy(move),
                occ_mask=torch.from_numpy(occ), focal=500.0, pp=(round(W / 2), round(H / 2)))


def make_clip(n_frames, H=480, W=854, seed=0):
    """Frames of one synthetic clip: the texture drifts with the flow's mean."""
    return [make_frame(H, W, seed=seed, shift=(-1.5 * k, 0.5 * k)) for k in range(n_frames)]


def init_splats(frame, num_points, seed=0, device="cpu", grown=False):
    """Raw (pre-activation) splat parameters exactly as SimpleGaussian
    .init_gaussians_from_image builds them (gflow/trainer.py:206-238).  ``grown`` rescales
    the splats so the projected sigma is log-normal around 2 px (a mid-optimisation
    footprint)."""
    from .geometry import pix2world
    from .sampling import complex_texture_sampling
    rng = np.random.default_rng(seed)
    H, W, _ = frame["image"].shape
    xys, depths, scales, rgbs, _ = complex_texture_sampling(frame["image"], frame["depth"], num_points, rng=rng)
    intr = torch.tensor([frame["focal"], frame["focal"], float(frame["pp"][0]), float(frame["pp"][1])])
    extr = torch.eye(4)[:3].contiguous()
    xys_t = torch.from_numpy(xys).float()
    depths = depths.float()
    xyz = pix2world(xys_t, depths, intr, extr)
    sc = scales * (depths / depths.min()).squeeze().numpy()
    sc = torch.clamp(torch.from_numpy(sc).float().unsqueeze(1).repeat(1, 3), max=1e-3)
    if grown:
        g = torch.Generator().manual_seed(seed + 1)
        sig_px = 2.0 * torch.exp(0.5 * torch.randn(xyz.shape[0], 1, generator=g))
        sc = (sig_px * depths / frame["focal"]).repeat(1, 3) * torch.exp(0.3 * torch.randn(xyz.shape[0], 3, generator=g))
    eps = 1e-15
    rgb = torch.logit(torch.clamp(torch.from_numpy(rgbs).float(), eps, 1 - eps))
    opacity = torch.logit(0.99 * torch.ones(xyz.shape[0], 1)) / 10.0
    g2 = torch.Generator().manual_seed(seed + 2)
    rotate = torch.nn.functional.normalize(torch.rand(xyz.shape[0], 4, generator=g2))
    out = dict(xyz=xyz, scale=torch.abs(sc), rotate=rotate, opacity=opacity, rgb=rgb, intr=intr, extr=extr)
    return {k: v.contiguous().to(device) for k, v in out.items()}
