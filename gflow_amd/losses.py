"""Loss terms of the GFlow iteration (gflow/trainer.py:452-530) on the device.

``image_loss`` is the fused photometric + SSIM + depth term: ONE library call
produces the loss value, the per-pixel error map used by densification and the
gradient wrt the rendered planes (two stencil kernels, gflow_amd/csrc/gfl_loss.hip).
The per-splat regularisers (var / scale / still / flow) are small gathers and stay
as torch expressions here; the fused trainer path folds them into its backward
kernel.
"""
import torch
import torch.nn.functional as F

from . import _lib as L


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render4, gt_rgb, gt_depth, keep, depth_ab, lambda_rgb, lambda_depth):
        lib = L.load()
        _, H, W = render4.shape
        dev = render4.device
        d_render = torch.empty_like(render4)
        err_px = torch.empty((H, W), dtype=torch.float32, device=dev)
        sums = torch.empty(8, dtype=torch.float32, device=dev)
        ws = L.scratch(lib.gfl_loss_workspace_bytes(W, H), dev)
        L.check(lib.gfl_loss_fwd_bwd(L.ptr(render4), L.ptr(gt_rgb), L.ptr(gt_depth), L.ptr(keep), L.ptr(depth_ab),
                                     lambda_rgb, lambda_depth, W, H, L.ptr(d_render), L.ptr(err_px), L.ptr(sums),
                                     L.ptr(ws), ws.numel(), L.stream()), "image_loss")
        hw = float(H * W)
        loss_rgb = sums[0] / hw + (1.0 - sums[1] / (3.0 * hw))
        loss_depth = sums[2] / hw
        loss = lambda_rgb * loss_rgb + lambda_depth * loss_depth
        ctx.save_for_backward(d_render, sums)
        ctx.mark_non_differentiable(err_px, loss_rgb, loss_depth)
        return loss, err_px, loss_rgb, loss_depth

    @staticmethod
    def backward(ctx, g_loss, _e, _r, _d):
        d_render, sums = ctx.saved_tensors
        return d_render * g_loss, None, None, None, sums[3:5] * g_loss, None, None


def image_loss(render4, gt_image, gt_depth, depth_ab, lambda_rgb=1.0, lambda_depth=0.0, move_mask=None):
    """render4 (4,H,W) = rgb planes + depth_map plane; gt_image (H,W,3); gt_depth (H,W,1)
    or (H,W) or None; depth_ab (2,) = [depth_a, depth_b] (trainer.py:145-146);
    move_mask (H,W) bool of MOVING pixels excluded in the camera-only phase
    (trainer.py:453-455,484).

    Returns (loss, err_px, loss_rgb, loss_depth) with
      loss_rgb   = mean(per-pixel mse) + 1 - SSIM          (trainer.py:459-462)
      loss_depth = mean((a D + b - gt)^2 / (a D + b + gt))  (trainer.py:479-485)
      loss       = lambda_rgb * loss_rgb + lambda_depth * loss_depth
      err_px     = per-pixel mse (H,W), the densification error map (trainer.py:459)."""
    L.need_device(render4, gt_image)
    if render4.dim() != 3 or render4.shape[0] != 4 or render4.dtype != torch.float32:
        raise RuntimeError("image_loss: render4 must be float32 (4,H,W)")
    _, H, W = render4.shape
    if tuple(gt_image.shape) != (H, W, 3):
        raise RuntimeError("image_loss: gt_image must be (H,W,3)")
    gt_rgb = gt_image.float().contiguous()
    gtd = None
    if lambda_depth != 0.0:
        if gt_depth is None or depth_ab is None:
            raise RuntimeError("image_loss: lambda_depth > 0 needs gt_depth and depth_ab")
        gtd = gt_depth.float().reshape(H, W).contiguous()
    if depth_ab is None:
        depth_ab = torch.tensor([1.0, 0.0], device=render4.device)
    keep = None
    if move_mask is not None:
        keep = (~move_mask.reshape(H, W).bool()).view(torch.uint8).contiguous()
    return _ImageLoss.apply(render4.contiguous(), gt_rgb, gtd, keep, depth_ab.float().contiguous(),
                            float(lambda_rgb), float(lambda_depth))


def ssim(img1, img2):
    """Mean SSIM of two (1,3,H,W) images through the fused kernel (value only)."""
    _, C, H, W = img1.shape
    r4 = torch.cat([img1[0], torch.zeros(1, H, W, device=img1.device)], dim=0)
    loss, _, loss_rgb, _ = image_loss(r4, img2[0].permute(1, 2, 0), None, None, 1.0, 0.0)
    mse = ((img1 - img2) ** 2).mean()
    return 1.0 - (loss_rgb - mse)


# ---- per-splat regularisers (trainer.py:490-530), plain torch on the device -------
def var_loss(scale):
    return torch.std(scale, dim=1).mean()


def scale_loss(scale, within_index, depth_point):
    return (torch.norm(scale[within_index], dim=1) * (1 / depth_point).squeeze()).mean()


def still_loss(xyz, last_xyz, last_still_mask):
    n = last_still_mask.shape[0]
    return torch.norm(xyz[:n][last_still_mask] - last_xyz[:n][last_still_mask], dim=1).mean()


def flow_loss(uv, last_uv, gt_flow, and_mask, last_num):
    pred = uv[:last_num][and_mask] - last_uv[and_mask]
    yy = last_uv[and_mask][:, 1].long()
    xx = last_uv[and_mask][:, 0].long()
    return F.mse_loss(pred, gt_flow[yy, xx])
