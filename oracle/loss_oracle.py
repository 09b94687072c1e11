"""CPU restatement of the loss terms, pose and geometry helpers on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Unlike the rasteriser, these
pieces of the reference are importable, so this file IS pinned: tests/test_oracle_golden.py
checks it against tests/golden/{ssim_small,ssim_480p,pix2world,colormap}.npz,
which tests/golden/make_golden.py captured from the reference modules themselves.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------- A10: gflow/utils/pytorch_ssim.py:7-37
def ssim_window(size=11, sigma=1.5, dtype=torch.float32):
    xs = torch.arange(size, dtype=torch.float64) - size // 2
    g = torch.exp(-(xs * xs) / (2.0 * sigma * sigma))
    # the reference builds the 1-D window in float32 (torch.Tensor of python
    # floats), normalises it, and takes the outer product in float32
    g = g.float()
    g = g / g.sum()
    return g.to(dtype)


def ssim(img1, img2, size=11):
    """img (1,C,H,W).  Mean SSIM with an 11x11 sigma=1.5 window, zero padding,
    C1=0.01^2, C2=0.03^2 (gflow/utils/pytorch_ssim.py:17-35)."""
    ch = img1.shape[1]
    w1 = ssim_window(size, 1.5, torch.float32)
    w2 = (w1.unsqueeze(1) @ w1.unsqueeze(0)).to(img1.dtype)
    win = w2.expand(ch, 1, size, size).contiguous()
    pad = size // 2

    def blur(x):
        return F.conv2d(x, win, padding=pad, groups=ch)

    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = blur(img1 * img1) - mu1_sq
    s2 = blur(img2 * img2) - mu2_sq
    s12 = blur(img1 * img2) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def rgb_loss(rendered_rgb, gt_image, move_mask=None):
    """gflow/trainer.py:452-462.  rendered (3,H,W), gt (H,W,3); optional (H,W)
    bool mask of MOVING pixels that are zeroed in both images (camera-only phase).
    Returns (loss_rgb, loss_rgb_pixel (H,W))."""
    if move_mask is not None:
        keep = ~move_mask
        rendered_rgb = rendered_rgb * keep.unsqueeze(0)
        gt_image = gt_image * keep.unsqueeze(-1)
    per_px = ((rendered_rgb.permute(1, 2, 0) - gt_image) ** 2).mean(dim=2)
    loss = per_px.mean() + (1 - ssim(rendered_rgb.unsqueeze(0), gt_image.permute(2, 0, 1).unsqueeze(0)))
    return loss, per_px


def depth_loss(depth_map, gt_depth, depth_a, depth_b, move_mask=None):
    """gflow/trainer.py:466,476-485.  depth_map (1,H,W), gt_depth (H,W,1)."""
    d = depth_a * depth_map.permute(1, 2, 0) + depth_b
    l = (d - gt_depth) ** 2 / (d + gt_depth)
    if move_mask is not None:
        l = l * (~move_mask).unsqueeze(-1)
    return l.mean()


def var_loss(scale):
    """gflow/trainer.py:491: mean over splats of the unbiased std of the 3 scales."""
    return torch.std(scale, dim=1).mean()


def scale_loss(scale, uv, depth, W, H, still_mask=None, camera_only=False):
    """gflow/trainer.py:495-502 with the index sets of :424-425 and :467-471.  The reference stores
    ``self.within_index = valid_uv_index`` -- the SAME tensor it then narrows in place to the still rows (camera-only
    stage) or the moving rows (joint stage) -- so ``scale[self.within_index]`` and ``depth_point = depth[valid_uv_index]``
    select one and the same set of rows.  scale (N,3) activated; uv (N,2), depth (N,1) of this iteration;
    still_mask (M,) bool with M <= N or None."""
    valid = (uv[:, 0] > 0) & (uv[:, 0] < W - 1) & (uv[:, 1] > 0) & (uv[:, 1] < H - 1)
    if still_mask is not None:
        valid = valid.clone()
        n = still_mask.shape[0]
        valid[:n] = (still_mask if camera_only else ~still_mask) & valid[:n]
    valid = valid.detach()
    return (torch.norm(scale[valid], dim=1) * (1.0 / depth[valid]).squeeze(-1)).mean()


def flow_loss(uv, last_uv, gt_flow, mask):
    """gflow/trainer.py:511-528.  uv (N,2) current; last_uv (M,2) with M<=N; gt_flow
    (H,W,2); mask (M,) bool.  GT flow is sampled at trunc(last_uv)."""
    m = mask
    pred = uv[:last_uv.shape[0]][m] - last_uv[m]
    yy = last_uv[m][:, 1].long()
    xx = last_uv[m][:, 0].long()
    return F.mse_loss(pred, gt_flow[yy, xx].to(pred.dtype))


def still_loss(xyz, last_xyz, last_still_mask):
    """gflow/trainer.py:505-507."""
    n = last_still_mask.shape[0]
    return torch.norm(xyz[:n][last_still_mask] - last_xyz[:n][last_still_mask], dim=1).mean()


# ------------------------------------------------- A2: gflow/trainer.py:115-121
def pose_to_extr(pose):
    """pose = [qx,qy,qz,qw, tx,ty,tz] (roma XYZW order, identity [0,0,0,1,0,0,0],
    gflow/trainer.py:41).  roma.RigidUnitQuat(Q,T).normalize().to_homogeneous()[:3]:
    q is L2-normalised, turned into a rotation matrix, T is the last column
    (signed_expm1 is the identity, gflow/utils/__init__.py:11-15)."""
    q = pose[:4]
    q = q / torch.linalg.norm(q)
    x, y, z, w = q[0], q[1], q[2], q[3]
    R = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)]),
        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)]),
        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]),
    ])
    return torch.cat([R, pose[4:7].unsqueeze(1)], dim=1)


# ------------------------------------------------- A5': gflow/utils/geometry.py:105-120
def pix2world(uv, depth, intr, extr):
    """uv (N,2) pixels, depth (N,1), intr (4,), extr (3,4) world->camera.  Uses
    intr[0] as the single focal for both axes, exactly like the reference
    (geometry.py:106 passes intr[0]) ."""
    rel = torch.cat([depth * (uv - intr[2:]) / intr[0], depth], dim=-1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=extr.dtype)
    cam2world = torch.linalg.inv(torch.cat([extr, bottom], dim=0))
    return rel @ cam2world[:3, :3].T + cam2world[:3, 3]
