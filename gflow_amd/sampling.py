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


def _probability(gray):
    gx, gy = _sobel(gray)
    mag = np.sqrt(gx ** 2 + gy ** 2)
    mag = mag + np.min(mag[mag > 0])                   # avoid zero probability
    return mag / np.sum(mag)


def complex_texture_sampling(gt_image, gt_depth, num_points=5000, device="cpu", mask=None, drop_to=None, rng=None):
    """gt_image (H,W,3) in [0,1], gt_depth (H,W,1).  Returns
    xys (n,2) int pixel coords (x,y), depths (n,1), scales_norm (n,), rgbs (n,3), gt_depth.
    ``rng`` (numpy Generator) makes the draw reproducible; the reference uses the
    global numpy state (complex_texture_sampling.py:24)."""
    rng = rng if rng is not None else np.random.default_rng()
    image = gt_image.detach().cpu().numpy() * 255
    gray = (0.299 * image[..., 0] + 0.587 * image[..., 1] + 0.114 * image[..., 2]).astype(np.float32)
    prob = _probability(gray)
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


def texture_probability_device(gt_image):
    """The sampling distribution of complex_texture_sampling (gradient magnitude of the grey image plus its smallest
    positive value, normalised) as an (H, W) float64 tensor ON THE IMAGE'S DEVICE: same arithmetic as the host version
    above (float32 grey image, float64 Sobel with OpenCV's default BORDER_REFLECT_101)."""
    import torch.nn.functional as F
    image = gt_image.detach().float() * 255
    gray = (0.299 * image[..., 0] + 0.587 * image[..., 1] + 0.114 * image[..., 2]).float()
    p = F.pad(gray.double()[None, None], (1, 1, 1, 1), mode="reflect")[0, 0]
    gx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    mag = torch.sqrt(gx * gx + gy * gy)
    floor = torch.where(mag > 0, mag, torch.full_like(mag, float("inf"))).min()
    mag = mag + floor
    return mag / mag.sum()


def complex_texture_sampling_device(gt_image, gt_depth, num_points, generator=None):
    """complex_texture_sampling without the host round trip (the image goes to the host, 410 000 probabilities through
    np.random.choice and five arrays back: ~25 ms per clip on the fit's critical path): the same distribution, drawn
    by inverse-CDF lookup on the device.  gt_image (H,W,3), gt_depth (H,W,1) on the device.  Returns DEVICE tensors
    xys (n,2) int64 (x, y), depths (n,1), scales_norm (n,) float64, rgbs (n,3) float32.  (No ``mask`` / ``drop_to``:
    those make the count data dependent; callers that pass them use the host version.)"""
    H, W = gt_image.shape[:2]
    prob = texture_probability_device(gt_image)
    flat = prob.flatten()
    cdf = torch.cumsum(flat, 0)
    u = torch.rand(num_points, generator=generator, device=gt_image.device, dtype=torch.float64) * cdf[-1]
    pts = torch.searchsorted(cdf, u, right=True).clamp_(max=flat.numel() - 1)
    ys, xs = pts // W, pts % W
    scales = 1.0 / flat[pts]
    scales_norm = scales * 100.0 / scales.sum()
    rgbs = (gt_image.detach().float() * 255)[ys, xs] / 255.0
    return torch.stack([xs, ys], dim=1), gt_depth[ys, xs], scales_norm, rgbs

