"""Gradient-magnitude importance sampling of the initial splats
(reference: gflow/utils/complex_texture_sampling.py:4-47).

cv2 is not available here, so the two cv2 calls are restated in numpy:
COLOR_RGB2GRAY (0.299 R + 0.587 G + 0.114 B) and the 3x3 Sobel with OpenCV's default
BORDER_REFLECT_101 border."""
import numpy as np
import torch


def _sobel(gray):
    p = np.pad(gray.astype(np.float64), 1, mode="reflect")          # reflect == REFLECT_101
    gx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return gx, gy


def complex_texture_sampling(gt_image, gt_depth, num_points=5000, device="cpu", mask=None, drop_to=None, rng=None):
    """gt_image (H,W,3) in [0,1], gt_depth (H,W,1).  Returns
    xys (n,2) int pixel coords (x,y), depths (n,1), scales_norm (n,), rgbs (n,3), gt_depth.
    ``rng`` (numpy Generator) makes the draw reproducible; the reference uses the
    global numpy state (complex_texture_sampling.py:24)."""
    rng = rng if rng is not None else np.random.default_rng()
    image = gt_image.detach().cpu().numpy() * 255
    gray = (0.299 * image[..., 0] + 0.587 * image[..., 1] + 0.114 * image[..., 2]).astype(np.float32)
    gx, gy = _sobel(gray)
    mag = np.sqrt(gx ** 2 + gy ** 2)
    mag = mag + np.min(mag[mag > 0])                   # avoid zero probability
    prob = mag / np.sum(mag)
    pts = rng.choice(np.arange(gray.size), size=num_points, p=prob.flatten())
    if mask is not None:
        flat = mask.squeeze().cpu().numpy().flatten().astype(bool)
        pts = pts[~flat[pts]]
    if drop_to is not None and len(pts) > drop_to:
        pts = rng.choice(pts, size=drop_to, replace=False)
    coords = np.unravel_index(pts, gray.shape)
    xys = np.array(coords).T[:, ::-1].copy()
    depths = gt_depth[coords]
    scales = 1 / prob[coords]
    scales_norm = scales * 100.0 / np.sum(scales)
    rgbs = image[coords] / 255.0
    return xys, depths, scales_norm, rgbs, gt_depth
