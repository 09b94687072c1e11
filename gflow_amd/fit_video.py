"""Video driver -- the counterpart of gflow/fit_video.py's frame loop, plus clip sharding.

``fit_clip`` keeps the reference's structure (fit_video.py:105-349): first-frame fit with
``iterations_first``; then for every later frame an optional camera-only fit
(``iterations_camera``, ``lr_camera_after``) followed by the full fit (``iterations_after``,
``lr_after``, occlusion-mask densification at iteration 0).  Inputs are in-memory frames (dicts
as produced by ``gflow_amd.synthetic.make_clip``: image, depth, flow, move_mask, occ_mask, focal,
pp); the reference's file readers and video writers are out of scope.

Multi-GPU: clips are independent, frames of one clip are not.  One process per GPU,
the clips dealt to the ranks by length (``shard``: longest first, each to the least loaded rank -- the
reference walks its sequences one after another, benchmark_multi.py:23-37), no data-path collective; at
the end ONE all-reduce(SUM) of a small metrics vector (which also carries every rank's own wall time in
a slot of its own) and ONE all-reduce(MAX) of the wall time (RCCL over xGMI on a node, tens of bytes:
pure latency).  ``--clips-per-gpu c``: a rank fits c of its clips AT THE SAME TIME on its GPU
(``fit_clips_concurrent``: one fit leaves the chip partly idle) -- the throughput mode of a node with more
clips than GPUs.

    python -m torch.distributed.run --nproc-per-node 8 -m gflow_amd.fit_video --clips 24 --frames 60 --clips-per-gpu 3
"""
import argparse
import json
import os
import time

import torch

# Canonical hyper-parameters: the README's example (README.md:85-110), which is what BASELINE.json's configs quote
# (60 000 splats, 500 / 300 iterations, 150 camera-only).  scripts/fit_video.sh:16-39 runs a different set
# (50 000 splats, lambda_depth 0.1, lambda_var 50, lambda_still 0, lr_after 4e-3, lr_camera_after 1e-3,
# densify_times_after 2, densify_occ_percent 0.5): pass those as ``cfg`` (SCRIPT_OVERRIDES) to reproduce that script.
# bench.py's STEP uses lambda_depth 0.1 so that the depth term is exercised at a weight where it matters; its clip fit
# uses these defaults.
SCRIPT_OVERRIDES = dict(num_points=50000, lr_after=4e-3, lr_camera_after=1e-3, densify_times_after=2,
                        densify_occ_percent=0.5, lambda_depth=0.1, lambda_var=50.0, lambda_still=0.0)
DEFAULTS = dict(num_points=60000, lr=4e-3, lr_camera=0.0, iterations_first=500, lr_after=1e-3, iterations_after=300,
                camera_first=True, lr_camera_after=5e-4, iterations_camera=150, densify_interval=150, densify_times=2,
                densify_interval_after=100, densify_times_after=1, densify_occ_percent=1.0, densify_err_thre=1e-2,
                densify_err_percent=1.0, lambda_rgb=1.0, lambda_depth=1e-4, lambda_var=10.0, lambda_still=10.0,
                lambda_flow=0.01, lambda_scale=0.0, background="black",
                # trajectories (fit_video.py:34-35 defaults 0 / 0; the README's and scripts/fit_video.sh's flags: 100 / 2):
                # with traj_num > 0 every frame ends with trainer.eval(traj_index, line_scale=0.5, point_scale=2., alpha=0.8)
                # and project_points of the seeds (fit_video.py:226-238, 335-349)
                traj_num=0, traj_offset=0)

METRIC_NAMES = ("psnr_sum", "frames", "iterations", "rasterisations", "clips", "splats_final")


def shard(n_items, rank, world, lengths=None):
    """Indices of the clips rank ``rank`` fits.  Clips cost what their frames cost (500 iterations for the first, 450 for
    every other one), and a job lasts as long as its slowest rank: longest-processing-time-first -- the clips in
    descending length (ties: lower index first), each to the rank with the least work so far (ties: lower rank) -- ends
    within one clip of the ideal.  Every rank computes the same assignment from the same lengths: nothing is exchanged.
    ``lengths=None``: all clips equally long (then this is i mod world)."""
    lengths = [1] * n_items if lengths is None else [int(v) for v in lengths]
    if len(lengths) != n_items:
        raise ValueError("shard: one length per clip")
    load = [0] * world
    mine = []
    for i in sorted(range(n_items), key=lambda j: (-lengths[j], j)):
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += lengths[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


def reduce_metrics(local, wall_seconds, dist=None, device="cpu", rank=0, world=1):
    """SUM of the metrics vector and MAX of the wall time over all ranks (identity without dist).  The vector's tail
    has one slot per rank: every rank writes its own wall time into its slot, so the one SUM also hands every rank all
    the ranks' times (``rank_wall_s``: the imbalance of the shard)."""
    own = [0.0] * world
    own[rank] = float(wall_seconds)
    vec = torch.tensor([float(local.get(k, 0.0)) for k in METRIC_NAMES] + own, dtype=torch.float64, device=device)
    wall = torch.tensor([float(wall_seconds)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    vals = vec.tolist()
    out = {k: float(v) for k, v in zip(METRIC_NAMES, vals)}
    out["wall_s"] = float(wall.item())
    out["rank_wall_s"] = [float(v) for v in vals[len(METRIC_NAMES):]]
    return out


_FIT_STREAMS = {}


def upload_clip(frames, device):
    """The clip's frames with every tensor resident on ``device`` (and ``occ_count``, the number of set pixels of the
    occlusion mask, counted here on the host).  torch copies pageable host memory to the device synchronously: handing
    the frames over one ``set_gt_*`` at a time, as the reference does after reading each frame from disk, stopped the
    host -- and drained the device's queue -- five times per frame.  A frame is ~10 MB (image, depth, flow, masks): a
    60-frame clip is 0.6 GB of 288."""
    out = []
    for fr in frames:
        d = dict(fr)
        occ = fr.get("occ_mask")
        if occ is not None and "occ_count" not in d:
            m = torch.as_tensor(occ).squeeze()
            if m.dim() == 3:
                m = m[..., 0] if m.shape[-1] in (1, 3) else m[0]
            d["occ_count"] = int((m > 0).sum())
        for k, v in fr.items():
            if isinstance(v, torch.Tensor) and k != "extr":
                d[k] = v.to(device, non_blocking=True)
        out.append(d)
    return out


def select_traj_seeds(tr, traj_num, traj_offset):
    """The splats whose trajectories are drawn (fit_video.py:163-211, ``grid_traj = True`` as the reference hard-codes it):
    pixels of a grid -- stride 50 inside the eroded still region, stride 15 inside the eroded moving region (``move_seg``, the
    hull mask of the first frame's fit; a 10 x 10 erosion, cv2.erode's default border: nothing eroded from the image's edge) --
    and for every grid pixel the splat whose projection (``last_uv``) lies closest, kept if its still / moving label is the
    region's.  Returns (indices: still seeds first, split_interval = their number or None without moving seeds).
    Without a hull mask (fewer than six moving splats: the reference would fail in cv2.erode) the plain rule of
    fit_video.py:165-167, every (N / traj_num)-th splat from ``traj_offset``.  Once per clip; the nearest-splat search runs on
    the device (the reference forms an N x Q x 2 array in numpy), one read-back of the few hundred indices."""
    import numpy as np
    from scipy.ndimage import minimum_filter
    n = tr.current_pts_num()
    interval = max(int(n / traj_num), 1)
    plain = list(range(n))[traj_offset::interval]
    move_seg = getattr(tr, "move_seg", None)
    if move_seg is None:
        return plain, None
    H, W = tr.H, tr.W
    erode = lambda m: minimum_filter(m, size=10, mode="constant", cval=255)       # kernel 10 x 10, anchor (5, 5): x-5 .. x+4
    move_er = erode(np.asarray(move_seg, dtype=np.uint8))
    still_er = erode((255 - np.asarray(move_seg, dtype=np.int32)).astype(np.uint8))
    s_still, s_move = 50, 15
    grid = lambda lo_i, hi_i, lo_j, hi_j, s: [(j, i) for i in range(lo_i, hi_i, s) for j in range(lo_j, hi_j, s)]
    sparse = [(j, i) for (j, i) in grid(s_still, H, s_still, W, s_still) if still_er[i, j]]
    if len(sparse) == 0:
        sparse = grid(s_still, H, s_still, W, s_still)
    dense = [(j, i) for (j, i) in grid(s_move, H - s_move, s_move, W - s_move, s_move) if move_er[i, j]]
    if len(sparse) == 0:                                 # (sic: the reference tests the SPARSE list again, :197)
        dense = grid(s_move, H - s_move, s_move, W - s_move, s_move)
    uv = tr.last_uv.detach().double()
    still = tr.still_mask.detach()

    def closest(points):                                 # utils.find_closest_point (tracking.py:24-26): argmin over the splats
        q = torch.tensor(points, dtype=torch.float64, device=uv.device)
        out = []
        for a in range(0, q.shape[0], 256):              # (chunks: N x 256 distances at a time)
            d = ((uv[:, None, :] - q[None, a:a + 256, :]) ** 2).sum(-1)
            out.append(d.argmin(dim=0))
        return torch.cat(out)

    if len(sparse) == 0:
        # (H or W <= 50: the stride-50 grid has no pixel at all -- the reference would index an empty array, :200-203; the
        #  plain rule is what it falls back to without a mask)
        return plain, None
    sp = closest(sparse)
    sp_still = sp[still[sp]]
    if len(dense):
        de = closest(dense)
        de_move = de[~still[de]]
        return torch.cat([sp_still, de_move]).tolist(), int(sp_still.shape[0])
    return sp_still.tolist(), None


def begin_frame(tr, frames, i, load_extr=True):
    """What the frame loop does before frame i >= 1 is fitted (fit_video.py:242-253): the new targets, the flow from frame
    i - 1 to i, the frame's camera pose if the sequence carries one."""
    fr = frames[i]
    tr.set_gt_image(fr["image"])
    tr.set_gt_depth(fr["depth"])
    tr.set_gt_flow(frames[i - 1]["flow"])                # flow from frame i-1 to i (fit_video.py:250)
    if load_extr and fr.get("extr") is not None:
        tr.load_camera(extr=fr["extr"])                  # fit_video.py:252-253


def stage_kwargs(c, frames, i, stage):
    """The keyword arguments of frame i's (>= 1) two ``train`` calls -- ``stage`` "camera": the camera-only stage
    (fit_video.py:256-278), "joint": splats and nothing else (:288-315) -- without the per-call ones (snapshot interval,
    loss weights every call shares).  One place, because bench.py pins its mid-clip step windows on exactly these stages."""
    fr = frames[i]
    if stage == "camera":
        return dict(iterations=c["iterations_camera"], lr_camera=c["lr_camera_after"], lambda_var=0.0, lambda_still=0.0,
                    lambda_flow=c["lambda_flow"], densify_interval=c["densify_interval"], densify_times=c["densify_times"],
                    camera_only=True, move_mask=fr["move_mask"])
    if stage == "joint":
        return dict(iterations=c["iterations_after"], lr=c["lr_after"], lr_camera=0.0, lambda_var=c["lambda_var"],
                    lambda_still=c["lambda_still"], lambda_flow=c["lambda_flow"], densify_interval=c["densify_interval_after"],
                    densify_times=c["densify_times_after"], mask=fr.get("occ_mask"), mask_count=fr.get("occ_count"),
                    move_mask=fr["move_mask"])
    raise ValueError(stage)


def fit_clip(frames, device, cfg=None, seed=0, snapshot_interval=0, fused=True, log=None, load_extr=True, keep=None,
             async_snapshots=None):
    """Fit one clip; returns the metrics dict of this clip (PSNR summed over its frames; with ``cfg["traj_num"]`` > 0 also
    ``"traj"``: the per-frame trajectory images and seed projections, host arrays -- what the reference's frame loop collects in
    ``frames_sequence_traj / frames_sequence_traj_upon / sequence_traj``).
    ``load_extr`` (default True, like the reference's flag): frames that carry a camera pose
    (``extr``, read from the sequence's camera files) load it before they are fitted
    (fit_video.py:115-116, :252-253).  ``keep``: a dict that receives the trainer (``keep["trainer"]``) and the
    per-frame PSNR as device scalars (``keep["psnr"]``) -- for tests and tools; with ``cfg["traj_num"]`` also the per-frame
    trajectory images and seed projections as they left for the host (``keep["traj"]``)."""
    dev_ = torch.device(device)
    g = fit_clip_steps(frames, device, cfg=cfg, seed=seed, snapshot_interval=snapshot_interval, fused=fused, log=log,
                       load_extr=load_extr, chunk=None, keep=keep,
                       **({} if async_snapshots is None else {"async_snapshots": async_snapshots}))

    def drive():
        try:
            while True:
                next(g)
        except StopIteration as e:
            return e.value

    if dev_.type == "cuda" and torch.cuda.current_stream(dev_) == torch.cuda.default_stream(dev_):
        # Never fit on the default stream: it is HIP's legacy NULL stream, which every other (blocking) stream
        # synchronises with -- the snapshot copies on the copy stream then run BETWEEN the fit's launches instead of
        # beside them (measured: the same 8-frame clip fit 0.95 s on a stream of its own, 1.09 s on the default stream).
        key = dev_.index
        if key not in _FIT_STREAMS:
            _FIT_STREAMS[key] = torch.cuda.Stream(device=dev_)
        fs = _FIT_STREAMS[key]
        fs.wait_stream(torch.cuda.current_stream(dev_))
        with torch.cuda.stream(fs):
            out = drive()
        torch.cuda.current_stream(dev_).wait_stream(fs)
        return out
    return drive()


def fit_clip_steps(frames, device, cfg=None, seed=0, snapshot_interval=0, fused=True, log=None, load_extr=True, chunk=None,
                   async_snapshots=True, keep=None, cu_count=0):
    """fit_clip as a generator: yields after every ``chunk`` iterations of a stage (None: never) and returns the metrics
    dict.  The caller owns the stream the work is enqueued on (fit_clips_concurrent gives every clip its own)."""
    from .trainer import SimpleGaussian
    c = dict(DEFAULTS)
    c.update(cfg or {})
    if any(isinstance(v, torch.Tensor) and not v.is_cuda for k, v in frames[0].items() if k != "extr"):
        frames = upload_clip(frames, device)         # (bench.py uploads before its clock starts: "inputs resident in HBM")
    f0 = frames[0]
    tr = SimpleGaussian(f0["image"], f0["depth"], num_points=c["num_points"], background=c["background"],
                        device=device, seed=seed, fused=fused)
    tr.async_snapshots = bool(async_snapshots)       # (trainer.py: snapshots composed beside the next iterations, or behind theirs)
    tr.cu_count = int(cu_count)                      # (the caller's stream is CU-masked: fit_clips_concurrent(partition=True))
    tr.load_camera(focal=f0["focal"], pp=f0["pp"])
    if load_extr and f0.get("extr") is not None:
        tr.load_camera(extr=f0["extr"])
    tr.init_gaussians_from_image(f0["image"], f0["depth"], num_points=c["num_points"])
    common = dict(lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"], lambda_scale=c["lambda_scale"],
                  densify_occ_percent=c["densify_occ_percent"], densify_err_thre=c["densify_err_thre"],
                  densify_err_percent=c["densify_err_percent"], snapshot_interval=snapshot_interval,
                  lazy_images=True,        # (the image lists train() returns are not read here: do not wait for them)
                  chunk=chunk)
    # first frame (fit_video.py:119-142)
    traj = int(c["traj_num"]) > 0
    yield from tr.train_steps(iterations=c["iterations_first"], lr=c["lr"], lr_camera=c["lr_camera"],
                              lambda_var=c["lambda_var"], densify_interval=c["densify_interval"],
                              densify_times=c["densify_times"], move_mask=f0["move_mask"],
                              move_seg=traj,   # (the seeds' grid needs the first frame's hull mask: host work, once per clip)
                              **common)
    traj_rec = []                                        # per frame: where the seeds are, the camera, the scene's rgb image

    def record_trajectories():
        # fit_video.py:226-238 / 335-349 call trainer.eval + project_points here, after every frame.  What they need of THIS
        # frame is recorded on the device -- the seeds' positions, the camera, the rendered scene (the fused kernels, now) --
        # and the trajectory overlays of all frames are drawn once the clip is fitted (draw_trajectories): a poly-line's point
        # count is data and the operator path sizes its lists on the host, i.e. two or three full stops of the host per frame
        # here, each with the next frame's set-up behind it (60-frame clip fit: 9.76 -> 9.24 frames/s when drawn per frame).
        with torch.no_grad():
            xyz_now = tr.get_attribute("xyz")[traj_index_t].detach().float().clone()
            extr_now = tr.get_extr().detach().clone()
            if tr.fused and tr.engine is not None and tr.engine.N == tr.current_pts_num():
                rgb_u8 = tr._render_scene_fused()[0].clone()
            else:
                from . import render as render_mod
                rgb_u8 = render_mod.render2img_device(render_mod.render_multiple(tr._input_group(detach=True), ["rgb"])["rgb"])
            tr.rasterisations_done += 1
        traj_rec.append((xyz_now, extr_now, rgb_u8))

    def draw_trajectories():
        from . import msplat
        from . import render as render_mod
        imgs, uvs = [], []
        with torch.no_grad():
            for xyz_now, extr_now, rgb_u8 in traj_rec:
                overlay = tr.eval_trajectories(xyz_now, extr_now, line_scale=0.5, point_scale=2.0, alpha=0.8,
                                               split_interval=split_interval)
                img_traj = render_mod.render2img_device(overlay)
                # screen blending as trainer.eval forms it (numpy: float64, truncation)
                upon = 1.0 - (1.0 - rgb_u8.double() / 255.0) * (1.0 - img_traj.double() / 255.0)
                imgs.append(torch.stack([img_traj, (upon * 255.0).to(torch.uint8)]))
                uvs.append(msplat.project_point(xyz_now, tr.intr, extr_now, tr.W, tr.H)[0])       # trainer.project_points
                if keep is not None:
                    keep.setdefault("traj_groups", []).append([x.clone() if isinstance(x, torch.Tensor) else x
                                                               for x in tr.last_traj_group])
        return torch.stack(imgs), torch.stack(uvs)

    if traj:
        traj_index, split_interval = select_traj_seeds(tr, int(c["traj_num"]), int(c["traj_offset"]))
        traj_index_t = torch.as_tensor(traj_index, device=tr.device).long()
        record_trajectories()
    # (PSNR stays on the device and is read ONCE at the end of the clip: a float() per frame drained the queue between
    #  two frames; with a log callback the caller asked for the numbers as they come)
    psnr_sum = tr.psnr().double()
    if keep is not None:
        keep["trainer"], keep["psnr"] = tr, [psnr_sum]
    if log:
        log(f"frame 0: psnr {float(psnr_sum):.2f} dB, splats {tr.current_pts_num()}")
    for i, fr in enumerate(frames[1:], start=1):
        begin_frame(tr, frames, i, load_extr)
        if c["camera_first"]:                            # fit_video.py:256-278
            yield from tr.train_steps(**stage_kwargs(c, frames, i, "camera"), **common)
        if c["iterations_after"] > 0:                    # fit_video.py:288-315
            yield from tr.train_steps(**stage_kwargs(c, frames, i, "joint"), **common)
        if traj:
            record_trajectories()
        p = tr.psnr()
        psnr_sum = psnr_sum + p.double()
        if keep is not None:
            keep["psnr"].append(p)
        if log:
            log(f"frame {i}: psnr {float(p):.2f} dB, splats {tr.current_pts_num()}")
    if tr.engine is not None:
        tr.engine.check_overflow()
    if traj:
        # one copy of every frame's two images and seed projections to the host, at the end of the clip (the reference copies
        # five images and the projections per frame, blocking: render2img, .cpu().numpy())
        imgs_d, uvs_d = draw_trajectories()
        if imgs_d.is_cuda:
            from .trainer import _PINNED
            block = _PINNED.take(imgs_d.numel())
            imgs_h = block[0][:imgs_d.numel()].view(imgs_d.shape)
            imgs_h.copy_(imgs_d, non_blocking=True)
            uvs_h = uvs_d.cpu()                          # (small; waits for the stream, and with it for the images above)
            traj_out = dict(images=_PINNED.hand_out(block, [imgs_h])[0], uv=uvs_h.numpy(), index=list(traj_index),
                            split_interval=split_interval)
            _PINNED.release(block)
        else:
            traj_out = dict(images=imgs_d.numpy(), uv=uvs_d.numpy(), index=list(traj_index), split_interval=split_interval)
        if keep is not None:
            keep["traj"] = traj_out
    out = dict(psnr_sum=float(psnr_sum), frames=len(frames), iterations=tr.iterations_done,
               rasterisations=tr.rasterisations_done, clips=1, splats_final=tr.current_pts_num(),
               # iterations that stepped nothing because a tile outgrew its reserved region, and were made up for
               void_iterations=getattr(tr.engine, "regions_outgrown", 0) if tr.engine is not None else 0)
    if traj:
        # what the reference appends per frame -- frames_sequence_traj, frames_sequence_traj_upon, sequence_traj
        # (fit_video.py:226-238, 335-349): images (frames, 2, H, W, 3) uint8 [trajectories alone, upon the render],
        # uv (frames, seeds, 2), index, split_interval.  Not a number: callers that sum the dicts skip it (NUMERIC_KEYS)
        out["traj"] = traj_out
    return out


# the keys of fit_clip's dict that are per-clip numbers (sums over clips make sense); "traj" is the trajectory output
NUMERIC_KEYS = ("psnr_sum", "frames", "iterations", "rasterisations", "clips", "splats_final", "void_iterations")


def fit_clips_concurrent(clips, device, cfg=None, seeds=None, snapshot_interval=0, chunk=32, partition=False):
    """Fit several clips AT THE SAME TIME on ONE device, in one host thread: every clip has its own trainer, engine and
    STREAM, and the clips take turns enqueueing ``chunk`` iterations each (fit_clip_steps), so their graph launches
    interleave on the device.  One fit leaves the chip partly idle -- its kernels are a chain of dependent launches,
    several of them latency bound with about one wave per SIMD, and the blend launches end with a tail of a few busy CUs
    -- so the launches of a second and third fit fill the gaps.  Clips are independent: this is the same
    sharding as one clip per GPU, applied inside a GPU.  A value read back by one fit (a densification event) stops
    the host only until THAT fit's stream has caught up; the others have their chunks queued meanwhile.
    (One host thread on purpose: with a thread per clip, graph captures of one thread and launches / allocations of
    another crashed inside the HIP runtime of ROCm 7.2 about once in four runs -- aborts in hipGraphDestroy, segmentation
    faults beside hipStreamEndCapture, silent exits.)  Returns the clips' metrics dicts in order; the caller times the call.
    ``partition``: every clip's stream is CU-MASKED to its own share of every XCD (_lib.cu_partition: 256 CUs / n, each share
    spanning all eight XCDs and their L2s) and its engines size their persistent blend grids and tile queues for that share
    (gfl_fit_state.cu_count) -- the clips then run SIDE BY SIDE instead of taking turns on every CU.  Results do not depend
    on it (the schedule never enters a result).  Measured in bench.py's ``clips_per_gpu`` table."""
    n = len(clips)
    seeds = list(range(n)) if seeds is None else seeds
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    cur = torch.cuda.current_stream(dev)
    shares = None
    if partition and n > 1:
        from . import _lib
        shares = _lib.cu_partition(n, dev)
        streams = [_lib.masked_stream(words, dev) for words, _ in shares]
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    for s in streams:
        s.wait_stream(cur)
    # (a lone fit takes its snapshots on a side stream; several fits already fill each other's gaps, and a side stream + shadow
    #  engine per clip cost them more than they give)
    gens = [fit_clip_steps(clips[i], dev, cfg, seed=seeds[i], snapshot_interval=snapshot_interval, chunk=chunk,
                           async_snapshots=n == 1, cu_count=shares[i][1] if shares else 0)
            for i in range(n)]
    results = [None] * n
    live = list(range(n))
    while live:
        for i in list(live):
            with torch.cuda.stream(streams[i]):
                try:
                    next(gens[i])
                except StopIteration as e:
                    results[i] = e.value
                    live.remove(i)
    for s in streams:
        cur.wait_stream(s)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="fit synthetic clips, one process per GPU")
    ap.add_argument("--clips", type=int, default=1)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--num_points", type=int, default=60000)
    ap.add_argument("--iterations_first", type=int, default=500)
    ap.add_argument("--iterations_after", type=int, default=300)
    ap.add_argument("--iterations_camera", type=int, default=150)
    ap.add_argument("--frames-list", default=None, help="comma-separated frame counts, one per synthetic clip (clips of "
                                                        "unequal length: the shard balances them)")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="clips a rank fits at the same time on its GPU (fit_clips_concurrent)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--sequence", action="append", default=None,
                    help="path of a prepared sequence folder (images + the reference's sibling folders, "
                         "gflow_amd/io.py); may be given several times, one clip each; default: synthetic clips")
    ap.add_argument("--resize", type=int, default=None, help="shorter image side after loading a --sequence")
    args = ap.parse_args(argv)
    from . import synthetic as S
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise RuntimeError("gflow_amd.fit_video needs a HIP device (there is no CPU rasteriser)")
    # (ranks share devices only on a box with fewer GPUs than ranks -- a functional run; RCCL refuses two ranks on
    # one device, so the two small metric all-reduces then go over gloo, as in bench.py)
    shared = world > n_dev
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    cfg = dict(num_points=args.num_points, iterations_first=args.iterations_first,
               iterations_after=args.iterations_after, iterations_camera=args.iterations_camera)
    local = {k: 0.0 for k in METRIC_NAMES}
    n_clips = len(args.sequence) if args.sequence else args.clips
    # how long is every clip?  (every rank must see the same numbers: they decide who fits what)
    if args.sequence:
        from . import io as gio
        lengths = [len(gio.sequence_paths(p)["img"]) for p in args.sequence]
    elif args.frames_list:
        lengths = [int(v) for v in args.frames_list.split(",")]
        if len(lengths) != n_clips:
            raise SystemExit("--frames-list needs one length per clip")
    else:
        lengths = [args.frames] * n_clips
    mine = shard(n_clips, rank, world, lengths)
    # reading / synthesising the clips is not part of the fit: do it before the clock starts
    t_load = time.perf_counter()
    clips = {}
    for ci in mine:
        if args.sequence:
            clips[ci] = gio.load_sequence(args.sequence[ci], resize=args.resize)
        else:
            clips[ci] = upload_clip(S.make_clip(lengths[ci], args.height, args.width, seed=ci, device=dev), dev)
    t_load = time.perf_counter() - t_load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c = max(1, args.clips_per_gpu)
    order = sorted(clips, key=lambda j: (-lengths[j], j))          # (clips of similar length share the GPU)
    for g0 in range(0, len(order), c):
        group = order[g0:g0 + c]
        if len(group) == 1:
            ci = group[0]
            res = [fit_clip(clips[ci], dev, cfg, seed=ci,
                            log=(lambda s, ci=ci: print(f"[rank {rank} clip {ci}] {s}")) if args.verbose else None)]
        else:
            res = fit_clips_concurrent([clips[ci] for ci in group], dev, cfg, seeds=group)
        for m in res:
            for k in METRIC_NAMES:
                local[k] += m[k]
    torch.cuda.synchronize()
    out = reduce_metrics(local, time.perf_counter() - t0, dist, torch.device("cpu") if (world > 1 and shared) else dev,
                         rank=rank, world=world)
    if rank == 0:
        out["frames_per_s"] = out["frames"] / out["wall_s"]
        out["iterations_per_s"] = out["iterations"] / out["wall_s"]
        out["psnr_mean_db"] = out["psnr_sum"] / max(out["frames"], 1.0)
        out["n_gpus"] = world
        out["clips_per_gpu"] = c
        out["load_s_rank0"] = t_load
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
