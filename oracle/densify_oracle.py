"""Pixel-error densification restated on the CPU (gflow/trainer.py:878-951).  TEST INFRASTRUCTURE ONLY.

The reference's own function cannot be imported (trainer.py imports msplat); what it calls for the back-projection,
``geometry.pix2world``, can, and ``loss_oracle.pix2world`` is pinned to it by tests/golden/pix2world.npz."""
import numpy as np
import torch

from . import loss_oracle as LO


def sampling_distribution(error_map, error_threshold=1e-3, mask=None):
    """trainer.py:880-900: (probability per pixel (H,W) float64, mask_ratio).  error_map (H,W) numpy."""
    err = np.asarray(error_map, dtype=np.float64)
    err = err + np.nanmin(err[err > 0])                       # uniform floor (:884)
    if mask is None:
        m = (err > error_threshold).squeeze()                 # (:887)
    else:
        m = np.asarray(mask).squeeze()
    m = m > 0
    err = err * m[:, :err.shape[1]]
    ratio = np.sum(m) / np.size(m)
    return err / np.sum(err), ratio


def densify_num(num_points, mask_ratio, percent):
    return int(num_points * mask_ratio * percent)             # (:901)


def new_splats(ys, xs, gt_image, gt_depth, intr, extr, num_points):
    """trainer.py:908-934 for the sampled pixel coordinates (ys, xs): raw xyz, scale, rotate, opacity, rgb."""
    ys = torch.as_tensor(ys).long()
    xs = torch.as_tensor(xs).long()
    xys = torch.stack([xs, ys], dim=1).float()                # (:911) unravel_index gives (y, x); [::-1] -> (x, y)
    depths = gt_depth[ys, xs].reshape(-1, 1).float()          # (:912)
    scales = np.ones(ys.shape[0]) * (1.0 / num_points)        # (:914)
    scales = scales * (depths.numpy() / depths.numpy().min()).squeeze(-1)       # (:916)
    xyz = LO.pix2world(xys, depths, intr, extr)               # (:922)
    scale = torch.abs(torch.from_numpy(scales).float().unsqueeze(1).repeat(1, 3))   # (:923,926) inverse of |x| is |x|
    rgbs = torch.clamp(gt_image[ys, xs].contiguous(), min=1e-15, max=1 - 1e-15)     # (:927-930)
    rgb = torch.logit(rgbs)
    rotate = torch.tensor([1.0, 0.0, 0.0, 0.0]).repeat(ys.shape[0], 1)              # (:932)
    opacity = torch.logit(0.99 * torch.ones(ys.shape[0], 1)) / 10.0                 # (:933-934, trainer.py:72-77)
    return dict(xyz=xyz, scale=scale, rotate=rotate, opacity=opacity, rgb=rgb)
