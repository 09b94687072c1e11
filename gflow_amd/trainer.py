"""Per-frame optimiser -- the counterpart of gflow/trainer.py :: SimpleGaussian.

Same public surface for the hot path (``__init__``, ``load_camera``,
``init_gaussians_from_image``, ``get_attribute``, ``get_extr``, ``add_optimizer``,
``train``, ``project_points``, ``save_checkpoint`` / ``load_checkpoint``,
``densify_by_pixels``) and the same optimisation semantics, including the quirks
SURVEY.md A13 lists (Adam + LinearLR rebuilt on every ``train`` call; densification
replaces the optimiser with a constant-lr Adam over the attributes only).

What is different on purpose (results unchanged):
  * rgb and depth are composited in ONE 4-channel blend instead of two;
  * ``depth_map_color`` / ``center`` are rendered only on the iterations whose
    snapshots are kept (every 10th, trainer.py:573-582), not on all of them;
  * no host synchronisation inside an iteration: loss scalars stay on the device
    (the reference calls .item() on every term for its progress bar), the colour map
    runs on the device, masked losses are evaluated as mask-weighted sums rather than
    boolean gathers;
  * densification samples on the device (torch.multinomial) instead of moving the
    error map to the host for np.random.choice.
Visualisation-only pieces (concave-hull segmentation, PNG/MP4 writers) are out of
scope.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import geometry, losses, msplat
from . import render as render_mod
from .optim import Adam, LinearLR
from .sampling import complex_texture_sampling


def pose_to_extr(pose):
    """pose [qx,qy,qz,qw,tx,ty,tz] (XYZW, identity [0,0,0,1,0,0,0]) -> (3,4) world->camera:
    what roma.RigidUnitQuat(Q,T).normalize().to_homogeneous()[:3] gives
    (trainer.py:115-121; signed_expm1 is the identity, utils/__init__.py:11-15)."""
    # R is linear in the ten products q_i q_j: ONE outer product and ONE (9 x 16) matrix-vector product instead of ~40
    # scalar kernels (this runs at every frame boundary and densification event of a fit, between two graph launches)
    q = pose[:4] / torch.linalg.norm(pose[:4])
    qq = (q.unsqueeze(1) * q.unsqueeze(0)).reshape(16)                  # [xx xy xz xw | yx yy yz yw | zx zy zz zw | wx wy wz ww]
    C, I9 = _quat_to_rot_constants(pose.device, pose.dtype)
    R = (I9 + C @ qq).reshape(3, 3)
    return torch.cat([R, pose[4:7].unsqueeze(1)], dim=1)


_Q2R = {}


def _quat_to_rot_constants(device, dtype):
    key = (str(device), dtype)
    if key not in _Q2R:
        xx, xy, xz, xw, yy, yz, yw, zz, zw = 0, 1, 2, 3, 5, 6, 7, 10, 11
        C = torch.zeros(9, 16, dtype=torch.float64)
        for row, terms in enumerate((
                ((yy, -2), (zz, -2)), ((xy, 2), (zw, -2)), ((xz, 2), (yw, 2)),
                ((xy, 2), (zw, 2)), ((xx, -2), (zz, -2)), ((yz, 2), (xw, -2)),
                ((xz, 2), (yw, -2)), ((yz, 2), (xw, 2)), ((xx, -2), (yy, -2)))):
            for col, val in terms:
                C[row, col] = val
        I9 = torch.eye(3, dtype=torch.float64).reshape(9)
        _Q2R[key] = (C.to(dtype).to(device), I9.to(dtype).to(device))
    return _Q2R[key]


def rotmat_to_unitquat_xyzw(R):
    """roma.rotmat_to_unitquat restated (XYZW, w >= 0 branch-free variant)."""
    m = R.double()
    t = m[0, 0] + m[1, 1] + m[2, 2]
    cands = torch.stack([
        torch.stack([m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1], 1 + t]),
        torch.stack([1 + m[0, 0] - m[1, 1] - m[2, 2], m[0, 1] + m[1, 0], m[0, 2] + m[2, 0], m[2, 1] - m[1, 2]]),
        torch.stack([m[0, 1] + m[1, 0], 1 - m[0, 0] + m[1, 1] - m[2, 2], m[1, 2] + m[2, 1], m[0, 2] - m[2, 0]]),
        torch.stack([m[0, 2] + m[2, 0], m[1, 2] + m[2, 1], 1 - m[0, 0] - m[1, 1] + m[2, 2], m[1, 0] - m[0, 1]]),
    ])
    best = torch.argmax(torch.stack([1 + t, 1 + m[0, 0] - m[1, 1] - m[2, 2], 1 - m[0, 0] + m[1, 1] - m[2, 2],
                                     1 - m[0, 0] - m[1, 1] + m[2, 2]]))
    q = cands[best]
    return (q / torch.linalg.norm(q)).to(R.dtype)


class _PinnedPool:
    """Page-locked staging blocks for the snapshots, re-used across train() calls.  Pinning costs ~0.15 ms per MB
    (hipHostMalloc of the 110-180 MB a stage's snapshots need: 16-43 ms, five times per later frame in a HIP API
    trace of a clip fit), so a block goes back to the pool as soon as the arrays handed out of it are gone."""

    MAX_BYTES = 4 << 30         # beyond this much page-locked memory the snapshots are handed out as pageable copies

    def __init__(self):
        import threading
        self.blocks = []                                # [uint8 pinned tensor, arrays still alive, event of the last copy INTO it]
        # re-entrant: the finalizers below take it too, and a garbage collection that runs them can start inside take()
        # (which allocates while it holds the lock) on the same thread
        self.lock = threading.RLock()                   # several fits may run in one process (fit_clips_concurrent)

    def total_bytes(self):
        return sum(b[0].numel() for b in self.blocks)

    def take(self, nbytes):
        """a free block of at least ``nbytes``; it counts as taken (one reference) until ``release``"""
        with self.lock:
            for b in self.blocks:
                if b[1] == 0 and b[0].numel() >= nbytes:
                    b[1] = 1
                    break
            else:
                step = 32 << 20
                b = [torch.empty((nbytes + step - 1) // step * step, dtype=torch.uint8, pin_memory=True), 1, None]
                self.blocks.append(b)
                return b
        # a caller that did not wait for its images (lazy_images) may have dropped them while the device-to-host copy
        # into this block was still queued: the next user must not be given the block before that copy has landed
        if b[2] is not None:
            b[2].synchronize()
            b[2] = None
        return b

    def copied(self, block, stream):
        """a device-to-host copy into ``block`` has just been queued on ``stream``"""
        ev = torch.cuda.Event()
        ev.record(stream)
        block[2] = ev

    def release(self, block):
        """drop the reference ``take`` left (after ``hold`` / ``hand_out`` have added theirs)"""
        with self.lock:
            block[1] -= 1

    def hold(self, block, owner):
        """the block stays taken while ``owner`` is alive"""
        import weakref

        def gone():
            with self.lock:
                block[1] -= 1
        with self.lock:
            block[1] += 1
        return weakref.finalize(owner, gone)             # call it to let go early

    def hand_out(self, block, tensors):
        """numpy views of ``tensors`` (views of the block); the block is free again when all of them are collected."""
        import weakref

        def gone():
            with self.lock:
                block[1] -= 1
        if self.total_bytes() > self.MAX_BYTES:
            # a caller that keeps every frame's snapshot lists (the reference's fit_video does, to write its videos)
            # would otherwise hold one 110-180 MB page-locked block per train() call: tens of GB over a 60-frame clip.
            # Pageable copies then -- taken only once the device has filled the block (lazy_images callers included)
            if block[2] is not None:
                block[2].synchronize()
            return [t.numpy().copy() for t in tensors]
        out = []
        for t in tensors:
            a = t.numpy()
            with self.lock:
                block[1] += 1
            weakref.finalize(a, gone)
            out.append(a)
        return out


_PINNED = _PinnedPool()
_COPY_STREAMS = {}


def _copy_stream(dev):
    """one side stream per device and FIT stream for the snapshot copies (creating a stream per train() call cost
    1.4 ms each; several fits on one device -- fit_clips_concurrent -- must not queue behind each other's copies)"""
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _COPY_STREAMS[key]


class _Stepper:
    """State of one ``train`` call; calling it runs one iteration, ``run(n)`` the next n."""

    def __call__(self):
        self.fn()

    def run(self, n):
        batch = getattr(self, "fn_batch", None)
        if batch is None:
            for _ in range(n):
                self.fn()
        else:
            batch(n)


def device_constant(values, device, dtype=torch.float32):
    """A small constant tensor WITHOUT a host-to-device copy: ``torch.tensor([...], device=)`` copies from pageable memory,
    which stops the host until everything queued on the device has run (7 ms apiece at the start of a fit, right behind the
    zero-filling of the engine's buffers: six of them were 4.6 % of an 8-frame clip fit).  Fills are just launches."""
    out = torch.zeros(len(values), dtype=dtype, device=device)
    for i, v in enumerate(values):
        if v != 0:
            out[i:i + 1].fill_(float(v))
    return out


def _within(uv, W, H):
    return (uv[:, 0] > 0) & (uv[:, 0] < W - 1) & (uv[:, 1] > 0) & (uv[:, 1] < H - 1)


class SimpleGaussian:
    def __init__(self, gt_image, gt_depth=None, gt_flow=None, num_points=100000, background="black",
                 device=None, log_dir=None, seed=None, fused=True):
        """``fused=True`` (default) runs each iteration as one call into the native library
        (gflow_amd/fused.py); ``fused=False`` composes the msplat-compatible autograd operators
        the way the reference does (slower, same results)."""
        self.fused = bool(fused)
        self.async_snapshots = True      # snapshots composed on a side stream from a copy of the forward's state (make_stepper)
        self.exact_snapshots = True      # iterations whose forward is looked at are never void or behind (make_stepper: one_iteration)
        self.cu_count = 0                # compute units the stream this trainer is driven on may use (0: the device): FitEngine(cu_count=)
        self.use_graph = True          # replay the fused iteration as a hipGraph when nothing else happens in it
        self.engine = None
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("gflow_amd.trainer needs a HIP device (there is no CPU rasteriser)")
        self.gt_image = gt_image.to(self.device)
        self.gt_depth = gt_depth.to(self.device) if gt_depth is not None else None
        self.gt_flow = gt_flow.to(self.device) if gt_flow is not None else None
        self.num_points = num_points
        H, W, _ = gt_image.shape
        self.H, self.W = H, W
        self.bg = {"black": 0.0, "white": 1.0, "cyan": 0.33}.get(background, 0.0)     # trainer.py:29-36
        fov = math.pi / 2.0
        fx = 0.5 * float(W) / math.tan(0.5 * fov)
        fy = 0.5 * float(H) / math.tan(0.5 * fov)
        self.intr = device_constant([fx, fy, W / 2.0, H / 2.0], self.device)
        self.pose = device_constant([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], self.device)
        self.rng = np.random.default_rng(seed)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(0 if seed is None else int(seed))
        N = int(num_points)
        self._activations = {
            "scale": torch.abs,
            "rotate": F.normalize,
            "opacity": lambda x: torch.sigmoid(x * 10.0),
            "rgb": torch.sigmoid,
        }
        self._activations_inv = {
            "scale": torch.abs,
            "rotate": F.normalize,
            "opacity": lambda x: torch.logit(x) / 10.0,
            "rgb": torch.logit,
        }
        rand = lambda *s: torch.rand(*s, device=self.device, generator=self.gen)
        self._attributes = {
            "xyz": rand(N, 3) * 2 - 1,
            "scale": rand(N, 3),
            "rotate": F.normalize(rand(N, 4)),
            "opacity": self._activations_inv["opacity"](0.99 * torch.ones(N, 1, device=self.device)),
            "rgb": rand(N, 3),
        }
        self.dir = log_dir
        if log_dir is not None:
            os.makedirs(log_dir, exist_ok=True)
        self.move_seg = None
        self.iterations_done = 0          # bookkeeping for throughput reports
        self.rasterisations_done = 0

    def set_gt_image(self, gt_image):
        self.gt_image = gt_image.to(self.device)

    def set_gt_depth(self, gt_depth):
        self.gt_depth = gt_depth.to(self.device)

    def set_gt_flow(self, gt_flow):
        self.gt_flow = gt_flow.to(self.device)

    # ------------------------------------------------------------------ camera
    def get_extr(self):
        return pose_to_extr(self.pose)

    def load_camera(self, focal=None, pp=None, extr=None, scale=None, show=False):
        if focal is not None:
            self.intr[:2].fill_(float(focal))
        if pp is not None:
            self.intr[2:3].fill_(float(pp[0]))
            self.intr[3:4].fill_(float(pp[1]))
        if extr is not None:
            extr = torch.as_tensor(extr, dtype=torch.float32, device=self.device)
            T = extr[:3, 3] * (scale if scale is not None else 1.0)
            pose = self.pose.detach().clone()
            pose[0:4] = rotmat_to_unitquat_xyzw(extr[:3, :3])
            pose[4:7] = T
            self.pose = pose
        if show:
            print("[camera] intr:", self.intr, "\n[camera] extr:\n", self.get_extr())

    # -------------------------------------------------------------- parameters
    def current_pts_num(self):
        return self._attributes["xyz"].shape[0]

    def get_attribute(self, name):
        if name not in self._attributes:
            raise ValueError(f"Attribute or activation for {name} is not VALID!")
        act = self._activations.get(name)
        return act(self._attributes[name]) if act is not None else self._attributes[name]

    def init_gaussians_from_image(self, gt_image, gt_depth=None, num_points=None, mask=None, drop_to=None):
        """trainer.py:206-238."""
        num_points = self.num_points if num_points is None else num_points
        self._engine_live = False                    # the attributes are replaced: no longer views of the engine's rows
        if mask is None and drop_to is None:
            # everything on the device (sampling.complex_texture_sampling_device): no image / depth / sample round trip
            from .sampling import complex_texture_sampling_device
            img = gt_image.to(self.device)
            self.gt_depth = gt_depth.float().to(self.device)
            xys, depths, scales, rgbs = complex_texture_sampling_device(img, self.gt_depth, num_points, generator=self.gen)
            n = xys.shape[0]
            xys, depths = xys.float(), depths.float()
            scales = (scales * (depths / depths.min()).squeeze(1).double()).float()
            rgbs = rgbs.float()
        else:
            xys, depths, scales, rgbs, gt_depth = complex_texture_sampling(
                gt_image, gt_depth.cpu(), num_points=num_points, mask=mask, drop_to=drop_to, rng=self.rng)
            n = xys.shape[0]
            xys = torch.from_numpy(xys).float().to(self.device)
            depths = depths.float().to(self.device)
            self.gt_depth = gt_depth.float().to(self.device)
            scales = torch.from_numpy(scales * (depths / depths.min()).squeeze().cpu().numpy()).float().to(self.device)
            rgbs = torch.from_numpy(rgbs).float().contiguous().to(self.device)
        self._attributes["xyz"] = geometry.pix2world(xys, depths, self.intr, self.get_extr().detach())
        scales = scales.unsqueeze(1).repeat(1, 3)
        self._attributes["scale"] = self._activations_inv["scale"](torch.clamp(scales, max=1e-3))
        rgbs = torch.clamp(rgbs.contiguous(), 1e-15, 1 - 1e-15)
        self._attributes["rgb"] = self._activations_inv["rgb"](rgbs)
        self._attributes["opacity"] = self._activations_inv["opacity"](0.99 * torch.ones(n, 1, device=self.device))
        self._attributes["rotate"] = F.normalize(torch.rand(n, 4, device=self.device, generator=self.gen))

    def add_optimizer(self, lr=1e-2, lr_camera=0.0, depth_invariant=True):
        """trainer.py:123-153: group "attributes" (lr), "extr" = pose (lr_camera), depth_a / depth_b
        (lr).  depth_a and depth_b live in one (2,) tensor here."""
        self.lr, self.lr_camera = lr, lr_camera
        self._engine_live = False
        for k in self._attributes:
            self._attributes[k] = nn.Parameter(self._attributes[k].detach().contiguous()).requires_grad_(True)
        self.pose = nn.Parameter(self.pose.detach().clone()).requires_grad_(True)
        self.depth_ab = nn.Parameter(device_constant([1.0, 0.0], self.device)).requires_grad_(depth_invariant)
        groups = [{"params": list(self._attributes.values()), "lr": lr, "name": "attributes"},
                  {"params": [self.pose], "lr": lr_camera, "name": "extr"}]
        if depth_invariant:
            groups.append({"params": [self.depth_ab], "lr": lr, "name": "depth_ab"})
        self.optimizer = Adam(groups)

    # ------------------------------------------------------------------ render
    def _input_group(self, sel=None, detach=False):
        g = []
        for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
            a = self.get_attribute(k)
            if detach:
                a = a.detach()
            if sel is not None:
                a = a[:sel.shape[0]][sel]
            g.append(a)
        return g + [self.intr, self.get_extr(), self.bg, self.W, self.H]

    def _render_rgbd(self, want_extras):
        """One pass of the rasteriser: uv, depth and the 4-plane render (rgb + depth_map);
        optionally the two snapshot-only images."""
        xyz, scale, rotate, opacity, rgb, intr, extr, bg, W, H = self._input_group()
        uv, depth = msplat.project_point(xyz, intr, extr, W, H)
        visible = depth != 0
        cov3d = msplat.compute_cov3d(scale, rotate, visible)
        conic, radius, tiles = msplat.ewa_project(xyz, cov3d, intr, extr, uv, W, H, visible)
        ids, tile_range = msplat.sort_gaussian(uv, depth, W, H, radius, tiles)
        self.last_K = ids.numel()
        render4 = msplat.alpha_blending(uv, conic, opacity, torch.cat([rgb, depth], dim=1), ids, tile_range, bg, W, H)
        extras = None
        if want_extras:
            with torch.no_grad():
                dc = render_mod.apply_float_colormap(depth.detach(), "turbo", non_zero=True)
                depth_color = msplat.alpha_blending(uv.detach(), conic.detach(), opacity.detach(), dc, ids,
                                                    tile_range, bg, W, H)
                unit = device_constant([1.0, 0.0, 1.0], self.device)
                center = msplat.alpha_blending(uv.detach(), torch.ones_like(conic) * unit,
                                               torch.ones_like(opacity.detach()), rgb.detach(), ids, tile_range, bg,
                                               W, H)
            extras = (depth_color, center)
        self.rasterisations_done += 1
        return uv, depth, render4, extras

    def init_mask_prompt_pts(self, mask_prompt, ckpt_name=None):
        """trainer.py:290-330 (fit_video.py:155-157 calls it once, after the first frame's fit, with the first frame's
        segmentation mask): ``self.mask_prompt_pts`` (N,) bool -- the splats that project inside the image AND onto a set
        pixel of ``mask_prompt`` (H, W).  From then on every joint ``train()`` leaves ``self.propagate_seg``, the hull of
        those splats' current projections (trainer.py:611-619).  With a log directory the prompt is written out as
        images_seg/propagate_mask_<ckpt_name>.png like the reference does."""
        with torch.no_grad():
            uv = render_mod.render_multiple(self._input_group(detach=True), ["uv", "center"])["uv"].detach()
        self.rasterisations_done += 1
        mask_prompt = torch.as_tensor(mask_prompt).to(self.device)
        within = _within(uv, self.W, self.H)
        yx = uv.long()
        under = mask_prompt[yx[:, 1].clamp(0, self.H - 1), yx[:, 0].clamp(0, self.W - 1)].bool()
        self.mask_prompt_pts = within & under
        if self.dir is not None and ckpt_name is not None:
            from PIL import Image
            os.makedirs(os.path.join(self.dir, "images_seg"), exist_ok=True)
            Image.fromarray((mask_prompt.detach().cpu().numpy() * 255).astype(np.uint8)).save(
                os.path.join(self.dir, "images_seg", f"propagate_mask_{ckpt_name}.png"))
        return self.mask_prompt_pts

    # ------------------------------------------------------------------- train
    def make_stepper(self, iterations=500, lr=1e-2, lr_camera=0.0, lambda_rgb=1.0, lambda_depth=0.0,
                     lambda_flow=0.0, lambda_var=0.0, lambda_still=0.0, lambda_scale=0.0, move_mask=None,
                     densify_interval=500, densify_times=1, mask=None, camera_only=False, densify_occ_percent=0.1,
                     densify_err_thre=1e-2, densify_err_percent=0.2, snapshot_interval=10, log_interval=0,
                     mask_count=None):
        """Set up one ``train`` call (pre-update, fresh Adam + LinearLR, trainer.py:347-384) and
        return a callable that runs ONE iteration of trainer.py:387-582 per call."""
        W, H, dev = self.W, self.H, self.device
        if move_mask is not None:
            move_mask = move_mask.to(dev).bool()

        # ---- pre-update: carry moving splats along the GT flow (trainer.py:348-376)
        if not camera_only and hasattr(self, "still_mask"):
            n_last = self.last_still_mask.shape[0]
            moving = ~self.last_still_mask
            uv_last = self.last_uv[:n_last]
            inside = _within(uv_last, W, H) & moving
            yx = uv_last.long()
            flow_at = self.gt_flow[yx[:, 1].clamp(0, H - 1), yx[:, 0].clamp(0, W - 1)]
            uv_new = uv_last + flow_at
            yn = uv_new[:, 1].long().clamp(0, H - 1)
            xn = uv_new[:, 0].long().clamp(0, W - 1)
            depth_new = self.gt_depth[yn, xn]
            xyz_new = geometry.pix2world(uv_new, depth_new.reshape(-1, 1), self.intr, self.get_extr().detach())
            xyz = self._attributes["xyz"].detach().clone()
            xyz[:n_last] = torch.where(inside.unsqueeze(1), xyz_new, xyz[:n_last])
            self._attributes["xyz"] = xyz

        if self.fused:
            return self._make_fused_stepper(
                iterations=iterations, lr=lr, lr_camera=lr_camera, lambda_rgb=lambda_rgb, lambda_depth=lambda_depth,
                lambda_flow=lambda_flow, lambda_var=lambda_var, lambda_still=lambda_still, lambda_scale=lambda_scale,
                move_mask=move_mask,
                densify_interval=densify_interval, densify_times=densify_times, mask=mask, camera_only=camera_only,
                densify_occ_percent=densify_occ_percent, densify_err_thre=densify_err_thre,
                densify_err_percent=densify_err_percent, snapshot_interval=snapshot_interval,
                log_interval=log_interval, mask_count=mask_count)

        self.add_optimizer(lr, lr_camera, depth_invariant=True)
        self.scheduler = LinearLR(self.optimizer, start_factor=1.0, end_factor=0.1, total_iters=iterations)
        later_frame = hasattr(self, "last_xyz")
        has_still = hasattr(self, "still_mask")
        n_still = self.still_mask.shape[0] if has_still else 0
        st = _Stepper()
        st.frames, st.frames_depth, st.frames_center, st.log = [], [], [], []
        st.iteration = 0
        st.move_mask, st.camera_only = move_mask, camera_only

        def one_iteration():
            iteration = st.iteration
            snap = bool(snapshot_interval) and iteration % snapshot_interval == 0
            uv, depth, render4, extras = self._render_rgbd(want_extras=snap)
            within = _within(uv.detach(), W, H)
            self.within_index = within
            mm = move_mask
            if hasattr(self, "still_mask_tentative") and camera_only:
                # moving-splat footprint joins the move mask (trainer.py:427-451)
                with torch.no_grad():
                    grp = self._input_group(sel=~self.still_mask_tentative, detach=True)
                    mrgb = render_mod.render_multiple(grp, ["rgb"])["rgb"]
                    self.rasterisations_done += 1
                    grey = 0.299 * mrgb[0] + 0.587 * mrgb[1] + 0.114 * mrgb[2]
                    mm = (grey > 0.0) | st.move_mask         # running union (trainer.py:451 rebinds move_mask)
                st.move_mask = mm
            loss, loss_rgb_pixel, l_rgb, l_depth = losses.image_loss(
                render4, self.gt_image, self.gt_depth if lambda_depth > 0 else None, self.depth_ab,
                lambda_rgb if lambda_rgb > 0 else 0.0, lambda_depth if lambda_depth > 0 else 0.0,
                mm if camera_only else None)
            terms = {"rgb": l_rgb, "depth": l_depth}

            valid = within
            if has_still:
                valid = within.clone()
                valid[:n_still] = (self.still_mask if camera_only else ~self.still_mask) & valid[:n_still]
            if lambda_var:
                l_var = losses.var_loss(self.get_attribute("scale"))
                loss = loss + lambda_var * l_var
                terms["var"] = l_var
            if lambda_scale:
                # trainer.py:495-502: the reference's within_index ALIASES valid_uv_index, which :467-471 narrow in
                # place, so scale and depth are both taken over `valid`
                l_scale = losses.scale_loss(self.get_attribute("scale"), valid, depth[valid])
                loss = loss + lambda_scale * l_scale
                terms["scale"] = l_scale
            if lambda_still and has_still:
                m = self.last_still_mask
                diff = torch.norm(self.get_attribute("xyz")[:m.shape[0]] - self.last_xyz[:m.shape[0]], dim=1)
                # (an empty selection gives the reference a NaN LOSS VALUE but finite gradients: the
                # clamp keeps both finite here)
                l_still = (diff * m).sum() / m.sum().clamp(min=1)
                loss = loss + lambda_still * l_still
                terms["still"] = l_still
            if lambda_flow and self.gt_flow is not None and hasattr(self, "last_uv"):
                and_mask = _within(self.last_uv, W, H)
                if has_still:
                    and_mask = and_mask.clone()
                    and_mask[:n_still] = (self.still_mask if camera_only else ~self.still_mask) & and_mask[:n_still]
                yx = self.last_uv.long()
                gt_f = self.gt_flow[yx[:, 1].clamp(0, H - 1), yx[:, 0].clamp(0, W - 1)]
                d = (uv[:self.last_num] - self.last_uv - gt_f) ** 2
                l_flow = (d * and_mask.unsqueeze(1)).sum() / (2.0 * and_mask.sum().clamp(min=1))   # mse over selected rows
                loss = loss + lambda_flow * l_flow
                terms["flow"] = l_flow

            self.optimizer.zero_grad()
            loss.backward()

            # ---- gradient control (trainer.py:535-551)
            if later_frame and self._attributes["rgb"].grad is not None:
                self._attributes["rgb"].grad.zero_()
            if has_still and self._attributes["xyz"].grad is not None:
                self._attributes["xyz"].grad[:n_still] *= (~self.still_mask).unsqueeze(1)
            if camera_only:
                for p in self._attributes.values():
                    if p.grad is not None:
                        p.grad.zero_()
            self.optimizer.step()
            self.scheduler.step()
            self.iterations_done += 1
            if log_interval and iteration % log_interval == 0:
                st.log.append({k: float(v.detach()) for k, v in terms.items()}
                              | {"total": float(loss.detach()), "it": iteration})

            # ---- densification (trainer.py:560-571)
            if not camera_only and iteration == 0 and later_frame and mask is not None:
                if mask.sum() > 0:
                    self.densify_by_pixels(torch.ones_like(loss_rgb_pixel), error_threshold=0.0,
                                           percent=densify_occ_percent, mask=mask)
            if (not camera_only and densify_interval and (iteration + 1) % densify_interval == 0
                    and (iteration + 1) // densify_interval <= densify_times):
                self.densify_by_pixels(loss_rgb_pixel, error_threshold=densify_err_thre, percent=densify_err_percent,
                                       mask=None)
            if snap:
                st.frames.append(render_mod.render2img_device(render4[:3]))
                st.frames_depth.append(render_mod.render2img_device(extras[0]))
                st.frames_center.append(render_mod.render2img_device(extras[1]))
            st.uv, st.depth, st.last_render = uv.detach(), depth.detach(), render4.detach()
            st.iteration += 1

        st.fn = one_iteration
        return st

    # ------------------------------------------------------- fused (native) path
    def _engine_for(self, n):
        from .fused import FitEngine
        if self.engine is None:
            self.engine = FitEngine(self.W, self.H, max(8 * int(self.num_points), 2 * n, 65536), self.device, bg=self.bg,
                                    cu_count=self.cu_count)
        self.engine.ensure_capacity(n)
        return self.engine

    def _pack_to_engine(self):
        """Copy the attribute tensors into the engine's packed rows and re-point
        ``_attributes`` at live views of them."""
        eng = self._engine_for(self.current_pts_num())
        if not (getattr(self, "_engine_live", False) and eng.N == self.current_pts_num()
                and self._attributes["xyz"].data_ptr() == eng.params.data_ptr()):
            eng.set_splats(self._attributes)         # (already live views of the engine's rows: nothing to copy)
        self._attributes = eng.views()
        self._engine_live = True
        return eng

    def _make_fused_stepper(self, iterations, lr, lr_camera, lambda_rgb, lambda_depth, lambda_flow, lambda_var,
                            lambda_still, lambda_scale, move_mask, densify_interval, densify_times, mask, camera_only,
                            densify_occ_percent, densify_err_thre, densify_err_percent, snapshot_interval,
                            log_interval, mask_count=None):
        """Same iteration as ``make_stepper`` but every step is ONE call into
        libgflow_hip (gfl_fit_iteration): no autograd graph, no torch kernels, no host sync."""
        W, H, dev = self.W, self.H, self.device
        self.lr, self.lr_camera = lr, lr_camera
        eng = self._pack_to_engine()
        eng._pend_event = None                     # (a watch the last stage queued and never read is not this stage's)
        eng.pose.copy_(self.pose.detach())
        eng.invalidate_regions()                   # (new pose, new frame's warp of the moving splats: the first iteration bins exactly)
        self.pose = eng.pose                       # live: get_extr() follows the optimised pose
        eng.depth_ab.zero_()                                         # trainer.py:145-146: (a, b) = (1, 0) on every train()
        eng.depth_ab[0:1].fill_(1.0)                                 # (fills, not torch.tensor(..., device=): that copy blocks)
        self.depth_ab = eng.depth_ab
        eng.intr.copy_(self.intr)
        eng.reset_optimizer()
        later_frame = hasattr(self, "last_xyz")
        has_still = hasattr(self, "still_mask")
        n = eng.N
        hp = eng.hp
        hp.bg = self.bg
        hp.lambda_rgb = lambda_rgb if lambda_rgb > 0 else 0.0
        hp.lambda_depth = lambda_depth if lambda_depth > 0 else 0.0
        hp.lambda_var, hp.lr, hp.lr_camera = lambda_var, lr, lr_camera
        hp.lambda_scale = float(lambda_scale or 0.0)
        hp.lr_end_factor, hp.total_iters = 0.1, iterations          # LinearLR(1.0 -> 0.1), trainer.py:384
        hp.freeze_rgb = 1 if later_frame else 0
        hp.freeze_all_splats = 1 if camera_only else 0
        hp.step_camera = 1
        hp.lambda_flow = hp.lambda_still = 0.0
        flow_target = flow_w = still_target = still_w = row_flags = None
        if has_still:
            row_flags = torch.zeros(n, dtype=torch.uint8, device=dev)
            # bit0: still (xyz frozen, trainer.py:543-546); bit1: the row has a label (scale term, :467-471)
            row_flags[:self.still_mask.shape[0]] = self.still_mask.to(torch.uint8) | 2
        if lambda_still and has_still:
            m = self.last_still_mask
            still_target = torch.zeros(n, 3, device=dev)
            still_target[:m.shape[0]] = self.last_xyz[:m.shape[0]]
            still_w = torch.zeros(n, device=dev)
            still_w[:m.shape[0]] = m.float() / m.sum().clamp(min=1)      # empty selection: weights 0, not 0/0
            hp.lambda_still = lambda_still
        if lambda_flow and self.gt_flow is not None and hasattr(self, "last_uv"):
            and_mask = _within(self.last_uv, W, H)
            if has_still:
                ns = self.still_mask.shape[0]
                and_mask = and_mask.clone()
                and_mask[:ns] = (self.still_mask if camera_only else ~self.still_mask) & and_mask[:ns]
            yx = self.last_uv.long()
            gt_f = self.gt_flow[yx[:, 1].clamp(0, H - 1), yx[:, 0].clamp(0, W - 1)]
            flow_target = torch.zeros(n, 2, device=dev)
            flow_target[:self.last_num] = self.last_uv + gt_f
            flow_w = torch.zeros(n, device=dev)
            flow_w[:self.last_num] = and_mask.float() / (2.0 * and_mask.sum().clamp(min=1))
            hp.lambda_flow = lambda_flow
        eng.set_regularisers(flow_target, flow_w, still_target, still_w, row_flags)
        eng.set_targets(self.gt_image, self.gt_depth if lambda_depth > 0 else None,
                        (~move_mask) if (camera_only and move_mask is not None) else None)

        st = _Stepper()
        st.frames, st.frames_depth, st.frames_center, st.log = [], [], [], []
        st.pin = st.copy_stream = st.pin_hold = st.ring = None
        st.iteration = 0
        st.move_mask, st.camera_only = move_mask, camera_only
        tentative = hasattr(self, "still_mask_tentative") and camera_only
        if tentative:
            # the footprint of the tentative moving splats joins the move mask in every iteration
            # (trainer.py:426-451): the library rebuilds ``keep`` inside its forward
            eng.set_footprint_mask(move_mask if move_mask is not None else torch.zeros(H, W, dtype=torch.bool),
                                   ~self.still_mask_tentative)

        def settle(last=None):
            """Iterations that stepped NOTHING since the last look (the pair lists overflowed: they are grown first; or a tile
            outgrew its reserved region: that one iteration was void) are run again, so that the fit is where one that never
            skipped an update is.  ``last``: what to run as the LAST of them instead of a plain iteration (an iteration
            whose forward the host looks at is taken again: one_iteration).  One blocking read of two words."""
            while True:
                k = eng.settle_overflow()
                if not k:
                    return
                for _ in range(k - 1 if last else k):
                    eng.iteration(use_graph=False)
                if last:
                    last()

        st.settle = settle
        st.unchecked = 0                 # iterations on reserved regions since the last look at the overflow words
        st.yielding = False              # run() returns early rather than wait for a look (train_steps with ``chunk``)

        def account():
            """run()'s watch, read: iterations that stepped nothing are run now, on the exact path (they cannot be void
            again); pair lists that overflowed or are more than half full (a looked-at iteration must not be the one that
            overflows them) go the blocking way -- grown, made up for."""
            got = eng.read_pending()
            if got is None:
                # (iterations run one at a time, st() instead of st.run(n): nothing was queued in between -- the blocking look)
                if st.unchecked:
                    settle()
                return
            code, skipped, pairs = got
            if code != 0 or 2 * pairs > eng.K_cap:
                if code == 0:
                    torch.cuda.current_stream().synchronize()
                    eng.grow_pairs()
                settle()
                return
            if skipped > 0:
                eng.overflow[1:2].zero_()
                eng.regions_outgrown = getattr(eng, "regions_outgrown", 0) + skipped
                for _ in range(skipped):
                    eng.iteration(use_graph=False, reserved=False)

        def one_iteration():
            iteration = st.iteration
            n_rendered = eng.N                       # rows this iteration projects (densification appends afterwards)
            snap = bool(snapshot_interval) and iteration % snapshot_interval == 0
            # Somebody LOOKS at this iteration's forward (snapshot, log entry, the error map of a densification).  An iteration
            # can step nothing (settle, above) -- its forward is then a render of truncated lists, and while it waits to be made
            # up for, the splats are a step behind the reference's at every later index (5 such iterations in the 27 050 of a
            # 60-frame clip, all in the first steps of the first frame).  So (a) a looked-at iteration bins on the exact path,
            # where no tile can outgrow a region, and so does the plain iteration in front of it; (b) BEFORE it is launched,
            # the iterations up to the one before that are accounted for (run(): watch_pending behind them, one more iteration
            # queued, then the look -- the host waits for work that is already done while the device runs that iteration; a
            # blocking look behind every looked-at iteration cost a clip fit 2 %, tools/ab_trainer_flag.py) and made up for on
            # the exact path.  The forward the host then looks at is the forward of the splats after exactly ``iteration``
            # optimiser steps, as in the reference (trainer.py:573-582).
            looked_at = not is_plain(iteration)
            if looked_at and self.exact_snapshots:
                account()
            if tentative:
                self.rasterisations_done += 1                # the reference's extra render of the moving set
            if snap:
                # The three images of THIS iteration's forward are composed on a stream of their own, from a COPY of what the
                # forward left behind (records, sorted ids, tile ranges, the render, the tile queues: 9 MB, one launch --
                # gfl_fit_snapshot_stage -- into a shadow engine), BESIDE the next iterations instead of between them.  Inside
                # the iteration's own graph the snapshot cost 115-140 us every tenth iteration, 4-5 % of a clip fit; two
                # independent chains of launches share the chip well (two clips on one GPU: 1.43x), a fork inside a graph
                # does not (docs/history.md section 7).
                # They stay on the DEVICE until the end of this train() call (a ring of uint8 images in HBM: 3.7 MB each at
                # 480p, 50 per call) and leave for page-locked memory in ONE copy then.  A copy to the host while the
                # iterations run holds up whatever kernel is running beside it for as long as it lasts -- 65 us, every
                # tenth iteration (tools/snapshot_timeline.sh, tools/d2h_probe.py: a chain of short kernels gets +57 us
                # per 74 us copy); at the end of the call it runs beside the host's set-up of the next stage, when the
                # device has little else to do.  (The reference blocks on three device-to-host copies right here.)
                k = len(st.frames)
                if st.pin is None or k >= st.pin.shape[0]:
                    # (k >= rows: a stepper that is run for more than ``iterations`` steps gets larger blocks)
                    n_snaps = max((iterations + snapshot_interval - 1) // snapshot_interval, 2 * k, 1)
                    block = _PINNED.take(n_snaps * 3 * H * W * 3)
                    pin = block[0][:n_snaps * 3 * H * W * 3].view(n_snaps, 3, H, W, 3)
                    hold = _PINNED.hold(block, st)                   # ... while this stepper lives
                    _PINNED.release(block)
                    ring = eng.snapshot_ring(n_snaps)                # (waits, on the stream, for the last call's copy)
                    if st.pin is not None:
                        if getattr(st, "snap_stream", None) is not None:
                            torch.cuda.current_stream().wait_stream(st.snap_stream)   # (the shadow engine writes the old ring)
                        ring[:k].copy_(st.ring[:k])
                        st.frames, st.frames_depth, st.frames_center = ([pin[j, c] for j in range(k)] for c in range(3))
                        st.pin_hold()
                    st.pin_block, st.pin, st.pin_hold, st.ring = block, pin, hold, ring
                    st.copy_stream = _copy_stream(dev)
                st.frames.append(st.pin[k, 0])
                st.frames_depth.append(st.pin[k, 1])
                st.frames_center.append(st.pin[k, 2])
            def launch():
                if snap and not self.async_snapshots:
                    # (several fits sharing the device -- fit_clips_concurrent -- already fill each other's gaps, and a side
                    #  stream and a shadow engine per clip cost them more than they give: 12.7 -> 10.9 frames/s with two clips.
                    #  There the snapshot stays behind the iteration, in the same graph launch; an elementwise kernel moves it
                    #  into the ring: ``copy_`` goes through the runtime's blit kernel, 47 us for these 3.7 MB)
                    imgs = eng.iteration(use_graph=self.use_graph, snapshot=True, reserved=False)
                    torch.bitwise_or(imgs, 0, out=st.ring[k])
                else:
                    # one call (or one hipGraph replay)
                    eng.iteration(use_graph=self.use_graph, reserved=None if not looked_at else False)
                    if snap:
                        self._snapshot_async(st.ring[k], n_rendered)
                        st.snap_stream = self._snap_stream

            launch()
            if looked_at and self.exact_snapshots == "blocking":
                settle(last=launch)                  # (the look BEHIND the iteration, and the iteration again: tests, A/B)
            if looked_at:
                st.unchecked = 0
            else:
                st.unchecked += 1
            self.rasterisations_done += 1
            self.iterations_done += 1
            rec_now = eng.rec                        # (densification may re-allocate the engine's buffers below)
            if log_interval and iteration % log_interval == 0:
                l_rgb, l_depth = eng.loss_terms()
                total = hp.lambda_rgb * l_rgb + hp.lambda_depth * l_depth
                entry = {"rgb": float(l_rgb), "depth": float(l_depth), "it": iteration}
                if lambda_var:
                    entry["var"] = float(losses.var_loss(torch.abs(eng.views()["scale"])))
                    total = total + lambda_var * entry["var"]
                entry["total"] = float(total)
                st.log.append(entry)

            # ---- densification (trainer.py:560-571)
            densified = False
            if not self.exact_snapshots and (not camera_only and densify_interval and (iteration + 1) % densify_interval == 0
                                             and (iteration + 1) // densify_interval <= densify_times):
                settle(last=launch)                  # (the error map below must be the scene's; this event reads back anyway)
            if not camera_only and iteration == 0 and later_frame and mask is not None:
                # (an empty mask appends nothing: densify_by_pixels's own single host read decides, there is no
                #  separate ``mask.sum() > 0`` read as in trainer.py:563)
                b, a = self.densify_by_pixels(torch.ones_like(eng.err_px), error_threshold=0.0,
                                              percent=densify_occ_percent, mask=mask, n_masked=mask_count)
                densified |= a > b
            if (not camera_only and densify_interval and (iteration + 1) % densify_interval == 0
                    and (iteration + 1) // densify_interval <= densify_times):
                b, a = self.densify_by_pixels(eng.err_px, error_threshold=densify_err_thre,
                                              percent=densify_err_percent, mask=None)
                densified |= a > b          # the reference swaps the optimiser only inside `if densify_num > 0` (:903-936)
            st.uv, st.depth, st.last_render = rec_now[:n_rendered, 0:2], rec_now[:n_rendered, 9:10], eng.render
            if densified:
                # trainer.py:941-951: the optimiser is replaced by Adam(attributes, lr): moments and step restart,
                # lr stays constant, pose / depth affine are no longer stepped.  Flags only: the new rows already
                # sit behind the old ones in the engine (densification_postfix), their regulariser weights and row
                # flags are the zeros the capacity-sized buffers were padded with.
                st.uv, st.depth = st.uv.clone(), st.depth.clone()
                eng.reset_optimizer(splats=True, camera=False)
                hp.total_iters = 0
                hp.step_camera = 0
            st.iteration += 1

        def is_plain(i):
            """nothing but the library's iteration happens in iteration i (no snapshot, log entry or densification)"""
            if snapshot_interval and i % snapshot_interval == 0:
                return False
            if log_interval and i % log_interval == 0:
                return False
            if not camera_only and i == 0 and later_frame and mask is not None:
                return False
            if (not camera_only and densify_interval and (i + 1) % densify_interval == 0
                    and (i + 1) // densify_interval <= densify_times):
                return False
            return True

        def run(n):
            """the next n iterations; runs of plain ones go into ONE graph launch, two or four at a time (2-6 us pass
            between two graph launches, tools/graph_gap.py)"""
            end = st.iteration + n
            while st.iteration < end:
                i = st.iteration
                if st.yielding and not is_plain(i) and self.exact_snapshots and not eng.pending_ready():
                    return                           # (the caller comes back: train_steps)
                if is_plain(i) and not is_plain(i + 1) and i + 1 < iterations and self.exact_snapshots:
                    # the plain iteration in front of a looked-at one: what ran before it is accounted for while IT runs
                    # (one_iteration: account) -- on the exact path, so that nothing is left unaccounted for.  Not the LAST
                    # iteration of the stage: index ``iterations`` is never run, so a watch queued there would be read by
                    # the next stage's first iteration -- after the end-of-stage settle() has already made up for the same
                    # void iterations (they were made up twice, under the next stage's hyper-parameters)
                    if st.unchecked:
                        eng.watch_pending()
                        st.unchecked = 0
                    n_rendered = eng.N
                    eng.iteration(use_graph=self.use_graph, reserved=False)
                    self.rasterisations_done += 2 if tentative else 1
                    self.iterations_done += 1
                    st.uv, st.depth, st.last_render = eng.rec[:n_rendered, 0:2], eng.rec[:n_rendered, 9:10], eng.render
                    st.iteration += 1
                    continue
                k = 0
                while (k < 4 and i + k < end and is_plain(i + k)
                       and (is_plain(i + k + 1) or i + k + 1 >= iterations or not self.exact_snapshots)):
                    k += 1
                if k < 2 or not self.use_graph:
                    one_iteration()
                    continue
                b = 4 if k == 4 else 2
                n_rendered = eng.N
                eng.iteration(use_graph=True, count=b)
                self.rasterisations_done += (2 * b) if tentative else b
                self.iterations_done += b
                st.uv, st.depth, st.last_render = eng.rec[:n_rendered, 0:2], eng.rec[:n_rendered, 9:10], eng.render
                st.iteration += b
                st.unchecked += b

        st.fn = one_iteration
        st.fn_batch = run
        return st

    def train(self, *a, **kw):
        """``train_steps`` run to the end (the generator exists so that several fits can take turns on one device,
        fit_video.fit_clips_concurrent); same arguments, returns what it returns."""
        g = self.train_steps(*a, **kw)
        try:
            while True:
                next(g)
        except StopIteration as e:
            return e.value

    def train_steps(self, iterations=500, save_ckpt=False, ckpt_name="ckpt", snapshot_interval=10, render_parts=True,
                    lazy_images=False, chunk=None, move_seg=False, **kw):
        """A generator: yields after every ``chunk`` iterations (None: never), returns train()'s tuple.
        One call = the optimisation of one frame (trainer.py:332-711); keyword arguments
        as ``make_stepper``.  Returns (frames, frames_center, frames_depth, still_rgb,
        still_center, move_rgb, move_center, move_seg) like the reference; the frame lists
        hold (H,W,3) uint8 snapshots taken every ``snapshot_interval`` iterations (0 = none).
        ``move_seg``: also build ``self.move_seg`` / ``self.move_seg_erode`` (trainer.py:604-609; off by default, see below).
        ``render_parts``: the four images of the still / moving splats the reference renders at the end of EVERY train()
        (trainer.py:632-677); False skips them (None in the tuple).  ``lazy_images``: return without waiting for the
        images -- they are views of page-locked memory that the device fills behind the queued work; read them after
        ``torch.cuda.synchronize()``.  (A caller that drops them, like fit_clip, saves one full stop of the host per call.)"""
        W, H, dev = self.W, self.H, self.device
        st = self.make_stepper(iterations=iterations, snapshot_interval=snapshot_interval, **kw)
        if chunk:
            # (several fits taking turns on one device from ONE host thread: a fit that would have to WAIT for its look at the
            #  overflow words hands the turn on instead -- run() returns early -- so that the others' queues do not run dry)
            st.yielding = True
            while st.iteration < iterations:
                st.run(min(int(chunk), iterations - st.iteration))
                yield
        else:
            st.run(iterations)
        self.train_log = st.log
        if self.fused and self.engine is not None:
            # Dropped (splat, tile) pairs must neither go unnoticed nor end the fit: the lists are grown and the iterations
            # that stepped nothing are run again (FitEngine.settle_overflow).  One read of two words per train() call.
            # (Round 6 measured what this full stop costs: the pause between two stages is 1.4-1.5 ms of DEVICE time with it and
            #  1.2 ms without -- an "exact tail" of 24 iterations behind a watch, the words read from that copy --, because the
            #  boundary's ~70 small torch kernels are a chain of launch latencies on the device whoever waits for whom; the tail's
            #  exact binning cost the 0.25 ms back.  tools/stage_times.py, tools/experiments/README.md.)
            st.settle()
            if getattr(st, "snap_stream", None) is not None:
                with torch.cuda.stream(st.snap_stream):
                    self._snap_aux.watch_overflow()
        camera_only, move_mask = st.camera_only, kw.get("move_mask")
        if move_mask is not None:
            move_mask = move_mask.to(dev).bool()

        # ---- post-update (trainer.py:588-625)
        if st.uv.shape[0] != self.current_pts_num():
            # splats were appended after the last render (densification on the final iteration; the
            # reference would fail on the shape mismatch at trainer.py:596): project them once more
            with torch.no_grad():
                if self.fused and self.engine is not None:
                    self.engine.forward()
                    st.uv, st.depth, st.last_render = self.engine.uv, self.engine.depth, self.engine.render
                else:
                    st.uv, st.depth, st.last_render, _ = self._render_rgbd(want_extras=False)
                self.rasterisations_done += 1
        uv_d, depth_d = st.uv.clone(), st.depth.clone()
        if not camera_only:
            within = _within(uv_d, W, H)
            yx = uv_d.long()
            labels = ~move_mask[yx[:, 1].clamp(0, H - 1), yx[:, 0].clamp(0, W - 1)]
            n_now = self.current_pts_num()
            still = torch.ones(n_now, dtype=torch.bool, device=dev)
            still[:uv_d.shape[0]] = torch.where(within, labels, still[:uv_d.shape[0]])
            self.still_mask = still
            self.still_mask_tentative = still.clone()
            if hasattr(self, "last_still_mask"):
                self.still_mask[:self.last_still_mask.shape[0]] = self.last_still_mask
            if move_seg:
                # trainer.py:604-609: the moving region as the smoothed concave hull of the moving splats' projections
                # (gflow_amd/hull.py).  Host work on a few thousand points -- the reference does it after every joint
                # train(); here only on request, because it reads uv back and stops the host (visualisation and
                # trajectory seeds use it, the optimisation does not).
                from .hull import FastConcaveHull2D
                from scipy.ndimage import minimum_filter
                sel = within & ~self.still_mask[:uv_d.shape[0]]
                pts = uv_d[sel].cpu().numpy()
                if pts.shape[0] > 5:
                    self.move_seg = (FastConcaveHull2D(pts).mask(W, H) * 255).astype(np.uint8)
                    # cv2.erode(move_seg, ones((20, 20))): minimum over x-10 .. x+9, nothing eroded from the border
                    self.move_seg_erode = minimum_filter(self.move_seg, size=20, mode="constant", cval=255)
            if getattr(self, "mask_prompt_pts", None) is not None:
                # trainer.py:611-619: the first frame's mask prompt, carried by the splats that lay under it: the smoothed
                # concave hull of where THOSE splats project now (the reference builds it after every joint train() once
                # init_mask_prompt_pts has been called; host work, like move_seg)
                from .hull import FastConcaveHull2D
                m = self.mask_prompt_pts
                p_uv = uv_d[:m.shape[0]][m]
                p_uv = p_uv[_within(p_uv, W, H)]
                if p_uv.shape[0] > 4:
                    self.propagate_seg = (FastConcaveHull2D(p_uv.cpu().numpy()).mask(W, H) * 255).astype(np.uint8)
            self.last_still_mask = self.still_mask.detach()
            self.last_uv = uv_d
            self.last_depth = depth_d
            self.last_xyz = self.get_attribute("xyz").detach()
            self.last_num = self.last_xyz.shape[0]

        still_rgb = still_center = move_rgb = move_center = None
        parts_pin = parts_block = None
        if hasattr(self, "still_mask") and render_parts:
            if self.fused and self.engine is not None and self.engine.N == self.still_mask.shape[0]:
                # both renders through the fused kernels on a second engine, the images converted on the device and on
                # their way to page-locked memory without stopping the host (four operator-path renders and four
                # blocking copies took ~3 ms per train() call: 4 % of a clip fit)
                parts_dev = self._render_parts_fused()
                parts_block = _PINNED.take(parts_dev.numel())
                parts_pin = parts_block[0][:parts_dev.numel()].view(parts_dev.shape)
                cs = _copy_stream(dev)
                cs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cs):
                    parts_pin.copy_(parts_dev, non_blocking=True)
                _PINNED.copied(parts_block, cs)
                parts_dev.record_stream(cs)
                if getattr(st, "copy_stream", None) is None:
                    st.copy_stream = cs
            else:
                with torch.no_grad():
                    o = render_mod.render_multiple(self._input_group(sel=self.still_mask, detach=True), ["rgb", "center"])
                    still_rgb, still_center = render_mod.render2img(o["rgb"]), render_mod.render2img(o["center"])
                    o = render_mod.render_multiple(self._input_group(sel=~self.still_mask, detach=True), ["rgb", "center"])
                    move_rgb, move_center = render_mod.render2img(o["rgb"]), render_mod.render2img(o["center"])
            self.rasterisations_done += 2
        self.last_render = st.last_render.clone()
        if save_ckpt:
            self.save_checkpoint(ckpt_name=ckpt_name)
        # the snapshots stayed on the device as uint8 images: ONE copy to the host here, not three
        # blocking copies every 10th iteration (trainer.py:573-582)
        if getattr(st, "pin", None) is not None and st.frames:
            k = len(st.frames)
            st.copy_stream.wait_stream(torch.cuda.current_stream())
            if getattr(st, "snap_stream", None) is not None:
                st.copy_stream.wait_stream(st.snap_stream)           # (the shadow engine's last images)
            with torch.cuda.stream(st.copy_stream):
                st.pin[:k].copy_(st.ring[:k], non_blocking=True)
                self.engine.snapshot_ring_copied()                   # (an event: the ring is free again after it)
            _PINNED.copied(st.pin_block, st.copy_stream)
            st.ring.record_stream(st.copy_stream)                    # (should the engine replace it by a larger one)
        if getattr(st, "copy_stream", None) is not None:
            if not lazy_images:
                st.copy_stream.synchronize()         # fused path: the images are already in pinned host memory
            to_host = lambda lst: _PINNED.hand_out(st.pin_block, lst) if lst else []
        else:
            to_host = lambda lst: [f for f in torch.stack(lst).cpu().numpy()] if lst else []
        if parts_pin is not None:
            still_rgb, still_center, move_rgb, move_center = _PINNED.hand_out(
                parts_block, [parts_pin[0, 0], parts_pin[0, 2], parts_pin[1, 0], parts_pin[1, 2]])
            _PINNED.release(parts_block)
        out = (to_host(st.frames), to_host(st.frames_center), to_host(st.frames_depth), still_rgb, still_center,
               move_rgb, move_center, self.move_seg)
        if getattr(st, "pin_hold", None) is not None:
            st.frames, st.frames_depth, st.frames_center, st.pin = [], [], [], None
            st.pin_hold()                            # (the stepper and its closure are a cycle: do not wait for the GC)
        return out

    def _snapshot_async(self, out, n):
        """The snapshot images of the forward the engine has just run (``n`` rows) into ``out`` ((3, H, W, 3) uint8 on the
        device), composed by a shadow engine on a side stream from a copy of that forward's state; returns at once.
        ``self._snap_stream`` is that stream."""
        import ctypes
        from .fused import FitEngine
        from . import _lib as L
        eng, dev = self.engine, self.device
        cur = torch.cuda.current_stream()
        aux = getattr(self, "_snap_aux", None)
        if aux is None or aux.cap < n or aux.K_cap < eng.K_cap:
            if aux is not None:
                # the engine grew.  The side stream may still be composing the previous snapshot from the old shadow's
                # buffers (allocated on the fit stream, used over there): let it finish before they are freed -- a rare
                # event, a wait of one snapshot -- and keep the SAME stream, so that everything that waits on
                # ``st.snap_stream`` (the ring's growth, the final copy to the host) still sees its last write
                self._snap_stream.synchronize()
            else:
                self._snap_stream = torch.cuda.Stream(device=dev)
            aux = self._snap_aux = FitEngine(self.W, self.H, max(eng.cap, n), dev, K_cap=eng.K_cap, bg=self.bg,
                                                  cu_count=self.cu_count)
            self._snap_done = None
        side = self._snap_stream
        if self._snap_done is not None:
            cur.wait_event(self._snap_done)          # the previous snapshot has read the shadow's buffers (ten iterations ago)
        aux.set_count(n)
        aux.hp.bg, aux.hp.nearest, aux.hp.extent = eng.hp.bg, eng.hp.nearest, eng.hp.extent
        L.check(eng.lib.gfl_fit_snapshot_stage(ctypes.byref(eng.state()), ctypes.byref(aux.state()), L.stream()),
                "snapshot stage")
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            aux.snapshot(out=out)
            self._snap_done = torch.cuda.Event()
            self._snap_done.record(side)

    def _render_scene_fused(self):
        """(3, H, W, 3) uint8 on the device: rgb, depth colour, centre blobs of the CURRENT splats and camera (what
        render_multiple(["rgb", "center", "depth_map_color"]) + render2img give, trainer.py:765-777) through the fused
        kernels on the second engine: one forward, one gfl_fit_snapshot; nothing is read back."""
        from .fused import FitEngine
        eng = self.engine
        n = eng.N
        aux = getattr(self, "_aux", None)
        if aux is None or aux.cap < n:
            aux = self._aux = FitEngine(self.W, self.H, max(eng.cap, n), self.device, bg=self.bg,
                                    cu_count=self.cu_count)
        aux.set_count(n)
        aux.pose.copy_(eng.pose)
        aux.intr.copy_(eng.intr)
        aux.hp.bg = self.bg
        aux.params[:n].copy_(eng.params[:n])
        aux.forward()
        out = aux.snapshot()
        aux.watch_overflow()
        return out

    def _render_parts_fused(self):
        """(2, 3, H, W, 3) uint8 on the device: [still splats, moving splats] x [rgb, depth colour, centre blobs] of the
        current state (trainer.py:632-677 renders rgb and centre of both sets).  A splat that is not in the set gets a raw
        opacity of -1000 -- sigmoid(-10^4) = 0 < 1/255, so the preprocess kernel never bins it -- instead of being
        gathered out: no boolean gather (that is a host read of the count), the depth order of the others is unchanged."""
        from .fused import FitEngine
        eng = self.engine
        n = eng.N
        aux = getattr(self, "_aux", None)
        if aux is None or aux.cap < n:
            aux = self._aux = FitEngine(self.W, self.H, max(eng.cap, n), self.device, bg=self.bg,
                                    cu_count=self.cu_count)
        aux.set_count(n)
        aux.pose.copy_(eng.pose)
        aux.intr.copy_(eng.intr)
        aux.hp.bg = self.bg
        out = []
        hidden = torch.full((), -1000.0, device=self.device)
        for sel in (self.still_mask, ~self.still_mask):
            aux.params[:n].copy_(eng.params[:n])
            aux.params[:n, 10] = torch.where(sel, eng.params[:n, 10], hidden)
            aux.forward()
            out.append(aux.snapshot())
        aux.watch_overflow()
        return torch.stack(out)

    # ------------------------------------------------------------ densification
    def densify_weights(self, error_map, error_threshold=1e-3, mask=None):
        """The sampling weights of trainer.py:880-897 on the device: (masked error map with the uniform floor, the
        boolean mask).  Nothing here reads back."""
        dev = self.device
        err = error_map.detach().float()
        inf = torch.full_like(err, float("inf"))
        pos_min = torch.where(err > 0, err, inf).min()
        # (no positive entry at all: np.nanmin of an empty selection -- the reference fails there; sample the mask
        # uniformly instead)
        err = err + torch.where(torch.isinf(pos_min), torch.ones_like(pos_min), pos_min)      # uniform floor (:884)
        if mask is None:
            m = err > error_threshold
        else:
            m = mask.detach().to(dev).squeeze()
            if m.dim() == 3:
                m = m[..., 0] if m.shape[-1] in (1, 3) else m[0]
            m = m > 0
        m = m[:, :err.shape[1]]
        return err * m, m

    def new_splats_at(self, ys, xs):
        """Raw attributes of the splats densification creates at the pixels (ys, xs) (trainer.py:910-934): at the
        ground-truth depth, isotropic scale depth / min(sampled depths) / num_points, the pixel's colour, identity
        rotation, opacity 0.99."""
        dev = self.device
        k = ys.shape[0]
        xys = torch.stack([xs, ys], dim=1).float()
        depths = self.gt_depth[ys, xs].reshape(-1, 1).float()
        scales = (depths / depths.min()).squeeze(1) * (1.0 / self.num_points)
        new_xyz = geometry.pix2world(xys, depths, self.intr, self.get_extr().detach())
        new_scale = torch.abs(scales.unsqueeze(1).repeat(1, 3))
        new_rgb = torch.logit(torch.clamp(self.gt_image[ys, xs].contiguous(), 1e-15, 1 - 1e-15))
        new_rot = torch.zeros(k, 4, device=dev)
        new_rot[:, 0] = 1.0
        new_op = torch.logit(0.99 * torch.ones(k, 1, device=dev)) / 10.0
        return new_xyz, new_scale, new_rot, new_op, new_rgb

    def densify_by_pixels(self, error_map, error_threshold=1e-3, percent=0.1, mask=None, n_masked=None):
        """trainer.py:878-939 on the device.  ONE host read per call -- the number of masked pixels, which fixes how
        many rows are appended (launch sizes and tensor shapes live on the host); the reference moves the whole error
        map to the host and samples with numpy.  ``n_masked``: that number, when the caller knows it already (the
        occlusion mask is an INPUT of the frame: fit_video counts it once when the clip is uploaded) -- no read at all."""
        W = self.W
        if n_masked == 0:
            return self.current_pts_num(), self.current_pts_num()
        err, m = self.densify_weights(error_map, error_threshold, mask)
        if n_masked is None:
            n_masked = int(m.sum())                                        # the host read
            if self.fused and self.engine is not None:
                self.engine.reap_graphs()            # (the device is idle right now: the one place per frame where retired graphs cost nothing to destroy)
        densify_num = int(self.num_points * (n_masked / m.numel()) * percent)     # float64 like numpy (:896-901)
        num_before = self.current_pts_num()
        if densify_num > 0:
            idx = self.sample_pixels(err, densify_num)
            self.last_densify_idx = idx
            self.densification_postfix(*self.new_splats_at(idx // W, idx % W))
        return num_before, self.current_pts_num()

    def sample_pixels(self, weights, count):
        """``count`` independent draws (with replacement) of flat pixel indices with probability weights / sum --
        np.random.choice(H*W, size, p) of trainer.py:905 -- by inverse-CDF lookup on the device (torch.multinomial
        over 4e5 categories took ~25 ms)."""
        cdf = torch.cumsum(weights.flatten().double(), 0)
        u = torch.rand(count, generator=self.gen, device=self.device, dtype=torch.float64) * cdf[-1]
        return torch.searchsorted(cdf, u, right=True).clamp_(max=cdf.numel() - 1)

    def densification_postfix(self, new_xyz, new_scale, new_rotate, new_opacity, new_rgb):
        """trainer.py:941-951 -- including the quirk that the new optimiser covers only the attributes, with a
        constant lr and fresh moments (the fused stepper turns that into flags, make_stepper).  On the fused path
        the rows are appended IN PLACE behind the engine's live rows (capacity-based buffers: no concatenation, no
        re-packing, the parameter views are simply re-cut)."""
        new = {"xyz": new_xyz, "scale": new_scale, "rotate": new_rotate, "opacity": new_opacity, "rgb": new_rgb}
        eng = self.engine
        if (self.fused and eng is not None and getattr(self, "_engine_live", False)
                and self._attributes["xyz"].data_ptr() == eng.params.data_ptr() and eng.N == self.current_pts_num()):
            from .fused import COLS
            n0, k = eng.N, new_xyz.shape[0]
            eng.ensure_capacity(n0 + k)
            rows = eng.params[n0:n0 + k]
            for name, (a, b) in COLS.items():
                rows[:, a:b] = new[name].detach().reshape(k, b - a)
            rows[:, 14:] = 0
            eng.set_count(n0 + k)
            self._attributes = eng.views()
            return
        for k in self._attributes:
            cat = torch.cat((self._attributes[k].detach(), new[k]), dim=0).contiguous()
            self._attributes[k] = nn.Parameter(cat).requires_grad_(True)
        self.optimizer = Adam(list(self._attributes.values()), lr=self.lr)

    # -------------------------------------------------------------- checkpoint
    def save_checkpoint(self, ckpt_name=None):
        """Same dict keys as trainer.py:252-272."""
        if self.dir is None:
            raise RuntimeError("save_checkpoint: this trainer was built without log_dir")
        ckpt = {
            # contiguous clones: on the fused path the attributes are column views of the engine's
            # [capacity][16] buffer, and torch.save would serialise the whole storage
            "attributes": {k: v.detach().clone().contiguous() for k, v in self._attributes.items()},
            "intr": self.intr,
            "extr": self.get_extr().detach().clone(),
            "still_mask": getattr(self, "still_mask", None),
            "move_seg": self.move_seg,
            "last_uv": getattr(self, "last_uv", None),
            "width": self.W,
            "height": self.H,
        }
        os.makedirs(os.path.join(self.dir, "ckpt"), exist_ok=True)
        self.checkpoint_path = os.path.join(self.dir, "ckpt", f"{ckpt_name or 'ckpt'}.tar")
        torch.save(ckpt, self.checkpoint_path)

    def load_checkpoint(self, checkpoint_path):
        ckpt = torch.load(checkpoint_path, map_location=self.device, weights_only=False)
        self._engine_live = False                    # the attributes are replaced: no longer views of the engine's rows
        self._attributes = {k: v.to(self.device) for k, v in ckpt["attributes"].items()}
        self.intr = ckpt["intr"].to(self.device)
        self.load_camera(extr=ckpt["extr"])
        for k in ("still_mask", "move_seg", "last_uv"):
            if ckpt.get(k) is not None:
                setattr(self, k, ckpt[k])

    # -------------------------------------------------------- trajectory render
    def eval(self, traj_index=None, line_scale=0.1, point_scale=0.3, alpha=0.5, split_interval=None):
        """trainer.py:713-811: render the current splats (rgb, center, depth_map_color) and the trajectories of the
        splats ``traj_index`` (poly-lines from their previous to their current positions, older segments fading by
        ``alpha`` per frame), plus the screen blend of both.  Returns five (H,W,3) uint8 images:
        (rgb, center, depth_colour, trajectories, rgb with the trajectories on top)."""
        traj_index = torch.as_tensor(traj_index, device=self.device).long()
        current_xyz = self.get_attribute("xyz")[traj_index].detach().float()
        with torch.no_grad():
            out_traj = self.eval_trajectories(current_xyz, self.get_extr().detach(), line_scale, point_scale, alpha, split_interval)
            out = render_mod.render_multiple(self._input_group(detach=True), ["rgb", "center", "depth_map_color"])
            self.rasterisations_done += 1
        out_img = render_mod.render2img(out["rgb"])
        out_img_center = render_mod.render2img(out["center"])
        out_img_depth = render_mod.render2img(out["depth_map_color"])
        out_img_traj = render_mod.render2img(out_traj)
        # screen blending
        result = 1 - (1 - np.array(out_img) / 255.0) * (1 - np.array(out_img_traj) / 255.0)
        return out_img, out_img_center, out_img_depth, out_img_traj, (result * 255).astype(np.uint8)

    def eval_trajectories(self, current_xyz, extr, line_scale=0.1, point_scale=0.3, alpha=0.5, split_interval=None):
        """The trajectory half of ``eval`` (trainer.py:716-762, 779-794) as a function of what it needs of a frame: where the
        tracked splats are (``current_xyz`` (n, 3)) and the frame's camera (``extr`` (3, 4)).  Advances the trajectory state
        (all poly-line points so far, their fading opacities and colours) by one frame and returns the overlay (3, H, W) float.
        fit_video.fit_clip records the two inputs after every frame and calls this for all frames once the clip is fitted:
        the poly-lines' point counts are data (a read-back per frame) and the operator path sizes its lists on the host."""
        from .color import apply_float_colormap
        from .trajectory import gen_line_set
        dev = self.device
        num_traj = current_xyz.shape[0]
        op_inv = self._activations_inv["opacity"]
        if not hasattr(self, "traj_xyz"):                                  # the first frame
            self.traj_xyz = current_xyz
            self.traj_scale = torch.ones((num_traj, 3), device=dev)
            self.traj_rotate = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).repeat(num_traj, 1)
            self.traj_opacity = op_inv(0.99 * torch.ones((num_traj, 1), device=dev))
            if split_interval is None or num_traj == split_interval:
                traj_rgb = torch.arange(0, 1, 1 / num_traj, device=dev).float().unsqueeze(1)
            else:
                still = torch.arange(0, 1, 1 / split_interval, device=dev).float().unsqueeze(1)
                move = torch.arange(0, 1, 1 / (num_traj - split_interval), device=dev).float().unsqueeze(1)
                traj_rgb = torch.cat([still, move], dim=0)
            traj_rgb = apply_float_colormap(traj_rgb, colormap="gist_rainbow")
            # The reference keeps logit(colour) and hands it to the rasteriser RAW (:735, :784-790): table entries that
            # are exactly 0 or 1 become -inf / +inf, i.e. "saturate wherever the blob has any weight".  A branch-free
            # compositor multiplies skipped splats by weight 0, and 0 * inf is NaN: keep the saturation, finite.
            self.traj_rgb = torch.nan_to_num(self._activations_inv["rgb"](traj_rgb), posinf=1e6, neginf=-1e6)
            self.last_traj_xyz = self.traj_xyz
            self.last_traj_rgb = self.traj_rgb
        else:                                                              # the following frames
            line_xyz, line_rgb = gen_line_set(self.last_traj_xyz, current_xyz, self.last_traj_rgb, device=dev)
            num_in_line = line_xyz.shape[0]
            self.traj_xyz = torch.cat([self.traj_xyz, line_xyz], dim=0)
            num_total = self.traj_xyz.shape[0]
            self.traj_scale = torch.ones((num_total, 3), device=dev) * 1e-6
            self.traj_rotate = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).repeat(num_total, 1)
            self.traj_opacity = torch.cat([self.traj_opacity * alpha,       # gradually fade out (raw values, :761)
                                           op_inv(0.99 * torch.ones((num_in_line, 1), device=dev))], dim=0)
            self.traj_rgb = torch.cat([self.traj_rgb, line_rgb], dim=0)
            self.last_traj_xyz = current_xyz
        # (the reference hands the RAW trajectory opacity / colour to the rasteriser, :784-790)
        traj_group = [self.traj_xyz, self.traj_scale, self.traj_rotate, self.traj_opacity, self.traj_rgb, self.intr, extr,
                      self.bg, self.W, self.H]
        self.last_traj_group = traj_group
        self.rasterisations_done += 1
        return render_mod.render_traj(traj_group, num_traj, line_scale, point_scale)

    def project_points(self, points):
        return msplat.project_point(points, self.intr, self.get_extr(), self.W, self.H)

    def psnr(self):
        return self.psnr_of(self.last_render)

    def psnr_of(self, render4):
        """PSNR of a rendered frame against the ground truth, on what the reference evaluates (benchmark.py:191-230): the
        SAVED image -- render2img's clamp, x 255, truncation to uint8 (render.py:158-166) -- read back as uint8 / 255."""
        img = torch.floor(torch.clamp(render4[:3].permute(1, 2, 0), 0.0, 1.0) * 255.0) / 255.0
        mse = ((img - self.gt_image) ** 2).mean()
        return -10.0 * torch.log10(mse)
