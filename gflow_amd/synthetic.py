"""Seeded synthetic frames and clips for benchmarks and tests.

There is no DAVIS data in the build container or on the GPU box, so the workload is
generated: an image that is a sum of random-phase sinusoids plus soft discs (so the
Sobel magnitude is non-degenerate), a depth plane with smooth bumps in [1,5], a flow
field, a disc-shaped moving region and a pinhole camera with fx=fy=500.  ``make_frame`` is ONE such frame (the
step benchmark's and the operator tests' scene); ``make_clip`` is a physically consistent clip of a rigid scene under a
translating camera with one independently moving object (``_Scene``)."""
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


class _Scene:
    """One rigid synthetic scene: a textured depth surface that does not move, a camera that translates
    by ``cam_step`` per frame along x, and ONE independently moving object -- a fronto-parallel textured disc in front of
    the surface.  World = the first frame's camera coordinates.  The surface is parametrised by the first frame's pixel
    coordinates (a, b): the point pix2world((a, b), D0(a, b)) carries the colour tex(a, b); both functions are analytic
    (the same sinusoids + soft discs and the same bumpy plane as ``make_frame``), so every frame is an exact resampling:
    a camera at (c, 0, 0) sees that point at u = a - fx c / D0(a, b), v = b (translation along x leaves Z and y / Z alone).
    Everything a frame carries follows from that geometry: depth, the flow to the NEXT frame (reprojection of the same
    point under the next camera; the disc's own motion on the disc), the move mask (the disc: what violates the epipolar
    constraint) and the occlusion mask (pixels whose point was hidden in the PREVIOUS frame: behind the disc or outside
    the image)."""

    def __init__(self, H, W, seed, focal=500.0, cam_step=0.01):
        self.H, self.W, self.f = H, W, float(focal)
        self.cam_step = float(cam_step)
        rng = np.random.default_rng(seed)
        self.waves = []
        for _ in range(6):                                      # (the draws of make_frame, in its order)
            fx, fy = rng.uniform(0.005, 0.05, 2)
            self.waves.append((fx, fy, rng.uniform(0, 2 * math.pi, 3), rng.uniform(0.05, 0.2, 3)))
        self.blobs = []
        for _ in range(20):
            cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(10, 60)
            self.blobs.append((cx, cy, r, rng.uniform(-0.4, 0.4, 3)))
        # the moving object: in front of everything (the surface is never nearer than 1.0), over ~10 % of the pixels
        self.obj_z = 0.9
        self.obj_r_px = math.sqrt(0.10 * H * W / math.pi)
        self.obj_c0 = (0.6 * W, 0.45 * H)                       # its centre in the first frame, pixels
        self.obj_vel = (0.008, -0.002)                          # its own velocity in the world, per frame (x, y)
        orng = np.random.default_rng(seed + 7919)
        self.obj_waves = [(orng.uniform(0.01, 0.06), orng.uniform(0.01, 0.06), orng.uniform(0, 2 * math.pi, 3),
                           orng.uniform(0.1, 0.25, 3)) for _ in range(3)]
        self.obj_base = orng.uniform(0.25, 0.75, 3)

    # ---- the static surface
    def tex(self, a, b):
        img = torch.zeros(a.shape + (3,), dtype=a.dtype, device=a.device)
        for fx, fy, ph, amp in self.waves:
            arg = 2 * math.pi * (fx * a + fy * b)
            for c in range(3):
                img[..., c] += amp[c] * torch.sin(arg + ph[c])
        for cx, cy, r, col in self.blobs:
            soft = torch.sigmoid(-(torch.sqrt((a - cx) ** 2 + (b - cy) ** 2) - r) / 2.0)
            img += soft.unsqueeze(-1) * torch.as_tensor(col, dtype=a.dtype, device=a.device)
        return torch.clamp(0.5 + img, 0.0, 1.0)

    def depth0(self, a, b):
        W = self.W
        d = 2.0 + 0.002 * (a - W / 2) + 0.5 * torch.sin(a / 90.0) * torch.cos(b / 70.0) + 0.3 * torch.sin(b / 40.0)
        return torch.clamp(d, 1.0, 5.0)

    def depth0_da(self, a, b):
        d = 2.0 + 0.002 * (a - self.W / 2) + 0.5 * torch.sin(a / 90.0) * torch.cos(b / 70.0) + 0.3 * torch.sin(b / 40.0)
        g = 0.002 + 0.5 / 90.0 * torch.cos(a / 90.0) * torch.cos(b / 70.0)
        return torch.where((d > 1.0) & (d < 5.0), g, torch.zeros_like(g))

    def surface_param(self, u, v, k):
        """a with a - fx c_k / D0(a, v) = u (Newton from the first-order guess; |d/da (fx c / D0)| < 1 for these clips)"""
        s = self.f * self.cam_step * k
        a = u + s / self.depth0(u, v)
        for _ in range(8):
            d = self.depth0(a, v)
            a = a - (a - s / d - u) / (1.0 + s * self.depth0_da(a, v) / (d * d))
        return a

    # ---- the moving object
    def obj_centre(self, k):
        """the disc's centre in frame k, pixels"""
        x0 = (self.obj_c0[0] - self.W / 2) * self.obj_z / self.f           # world position in the first frame
        y0 = (self.obj_c0[1] - self.H / 2) * self.obj_z / self.f
        x, y = x0 + k * self.obj_vel[0], y0 + k * self.obj_vel[1]
        return (self.f * (x - self.cam_step * k) / self.obj_z + self.W / 2, self.f * y / self.obj_z + self.H / 2)

    def obj_tex(self, ox, oy):
        img = torch.zeros(ox.shape + (3,), dtype=ox.dtype, device=ox.device)
        for fx, fy, ph, amp in self.obj_waves:
            arg = 2 * math.pi * (fx * ox + fy * oy)
            for c in range(3):
                img[..., c] += amp[c] * torch.sin(arg + ph[c])
        return torch.clamp(img + torch.as_tensor(self.obj_base, dtype=ox.dtype, device=ox.device), 0.0, 1.0)

    def frame(self, k, device="cpu"):
        H, W, f = self.H, self.W, self.f
        dt = dict(dtype=torch.float64, device=device)
        v, u = torch.meshgrid(torch.arange(H, **dt), torch.arange(W, **dt), indexing="ij")
        a = self.surface_param(u, v, k)
        d_bg = self.depth0(a, v)
        ocx, ocy = self.obj_centre(k)
        dist = torch.sqrt((u - ocx) ** 2 + (v - ocy) ** 2)
        move = dist < self.obj_r_px
        cover = torch.clamp(0.5 + (self.obj_r_px - dist), 0.0, 1.0).unsqueeze(-1)      # one-pixel soft rim (colour only)
        img = (1.0 - cover) * self.tex(a, v) + cover * self.obj_tex(u - ocx, v - ocy)
        depth = torch.where(move, torch.full_like(d_bg, self.obj_z), d_bg)
        # flow to frame k + 1: the same surface point under the next camera; the disc's own step on the disc
        ncx, ncy = self.obj_centre(k + 1)
        flow_bg = torch.stack([-(f * self.cam_step) / d_bg, torch.zeros_like(d_bg)], dim=-1)
        flow_obj = torch.tensor([ncx - ocx, ncy - ocy], **dt).expand(H, W, 2)
        flow = torch.where(move.unsqueeze(-1), flow_obj, flow_bg)
        # occlusion: background pixels whose point was behind the disc, or outside the image, in frame k - 1
        if k > 0:
            up = a - f * self.cam_step * (k - 1) / d_bg
            pcx, pcy = self.obj_centre(k - 1)
            hidden = torch.sqrt((up - pcx) ** 2 + (v - pcy) ** 2) < self.obj_r_px
            outside = (up < 0) | (up > W - 1)
            occ = ~move & (hidden | outside)
        else:
            occ = torch.zeros_like(move)
        extr = torch.eye(4, dtype=torch.float32)[:3].clone()
        extr[0, 3] = -self.cam_step * k
        return dict(image=img.float(), depth=depth.float().unsqueeze(-1), flow=flow.float(), move_mask=move, occ_mask=occ,
                    focal=f, pp=(round(W / 2), round(H / 2)), extr_gt=extr)


def make_clip(n_frames, H=480, W=854, seed=0, device="cpu", cam_step=0.01):
    """Frames of one rigid synthetic clip (``_Scene``): frame k is the scene seen from a camera at (k cam_step, 0, 0).
    ``extr_gt`` is that camera (world -> camera, (3, 4)); it is NOT loaded by fit_clip (the key the loader looks at is
    ``extr``): the camera-only stage has to find it.  ``device``: where the frames are synthesised (float64 torch ops)
    and left."""
    sc = _Scene(H, W, seed, cam_step=cam_step)
    return [sc.frame(k, device) for k in range(n_frames)]


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
