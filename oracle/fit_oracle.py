"""One iteration of the GFlow first-frame fit on the CPU (gflow/trainer.py:387-558),
built from the oracle operators.  TEST INFRASTRUCTURE ONLY: used by the parity tests
and timed as the ``cpu_baseline`` ("port": the reference has no CPU rasteriser of its
own) next to the HIP path in bench.py."""
import torch
import torch.nn.functional as F

from . import loss_oracle as LO
from . import msplat_oracle as MO


def activate(raw):
    """trainer.py:64-69,243-250."""
    return [raw["xyz"], torch.abs(raw["scale"]), F.normalize(raw["rotate"]), torch.sigmoid(10.0 * raw["opacity"]),
            torch.sigmoid(raw["rgb"])]


def fit_loss(raw, pose, depth_ab, intr, frame, bg, lambda_rgb, lambda_depth, lambda_var, timers=None):
    """Forward of one iteration: render rgb + depth_map, losses as trainer.py:452-493.
    Returns (loss, dict of pieces).  ``timers`` (dict) accumulates seconds per phase."""
    import time
    t0 = time.perf_counter()
    H, W, _ = frame["image"].shape
    xyz, scale, rot, op, rgb = activate(raw)
    extr = LO.pose_to_extr(pose)
    uv, depth = MO.project_point(xyz, intr, extr, W, H)
    vis = depth != 0
    cov = MO.compute_cov3d(scale, rot, vis)
    conic, radius, tiles = MO.ewa_project(xyz, cov, intr, extr, uv, W, H, vis)
    ids, tr = MO.sort_gaussian(uv, depth, W, H, radius, tiles)
    r4 = MO.alpha_blending(uv, conic, op, torch.cat([rgb, depth], dim=1), ids, tr, bg, W, H)
    t1 = time.perf_counter()
    l_rgb, err_px = LO.rgb_loss(r4[:3], frame["image"])
    loss = lambda_rgb * l_rgb
    l_depth = LO.depth_loss(r4[3:4], frame["depth"], depth_ab[0], depth_ab[1])
    if lambda_depth > 0:
        loss = loss + lambda_depth * l_depth
    l_var = LO.var_loss(scale)
    if lambda_var:
        loss = loss + lambda_var * l_var
    if timers is not None:
        timers["render_fwd"] = timers.get("render_fwd", 0.0) + (t1 - t0)
        timers["loss_fwd"] = timers.get("loss_fwd", 0.0) + (time.perf_counter() - t1)
    return loss, dict(render4=r4, uv=uv, depth=depth, err_px=err_px, l_rgb=l_rgb, l_depth=l_depth, l_var=l_var,
                      K=int(ids.numel()))


class OracleFit:
    """Raw parameters + torch.optim.Adam + LinearLR exactly as trainer.py:123-153,383-384."""

    def __init__(self, raw, intr, frame, lr, iterations, bg=0.0, lambda_rgb=1.0, lambda_depth=0.0, lambda_var=0.0,
                 lr_camera=0.0):
        self.raw = {k: raw[k].detach().clone().requires_grad_(True) for k in ("xyz", "scale", "rotate", "opacity", "rgb")}
        self.pose = torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], requires_grad=True)
        self.depth_ab = torch.tensor([1.0, 0.0], requires_grad=True)
        self.intr, self.frame, self.bg = intr, frame, bg
        self.lams = (lambda_rgb, lambda_depth, lambda_var)
        self.opt = torch.optim.Adam([{"params": list(self.raw.values()), "lr": lr},
                                     {"params": [self.pose], "lr": lr_camera},
                                     {"params": [self.depth_ab], "lr": lr}])
        self.sched = torch.optim.lr_scheduler.LinearLR(self.opt, start_factor=1.0, end_factor=0.1, total_iters=iterations)

    def step(self, timers=None):
        import time
        loss, info = fit_loss(self.raw, self.pose, self.depth_ab, self.intr, self.frame, self.bg, *self.lams,
                              timers=timers)
        t0 = time.perf_counter()
        self.opt.zero_grad()
        loss.backward()
        t1 = time.perf_counter()
        self.opt.step()
        self.sched.step()
        if timers is not None:
            timers["backward"] = timers.get("backward", 0.0) + (t1 - t0)
            timers["adam"] = timers.get("adam", 0.0) + (time.perf_counter() - t1)
        return loss.item(), info
