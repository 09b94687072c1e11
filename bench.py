"""Benchmark of the hot path at BASELINE.json configs[1]: 480x854, 60 000 splats, synthetic data.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: bench.py re-executes itself under torch.distributed.run, one rank per GPU)

Two measurements per rank, both with the inputs resident in HBM and bracketed by a barrier and a
device synchronisation:
  * the step: K optimisation iterations of the first-frame fit (rasterise forward, photometric + SSIM +
    depth + var loss, rasterise backward, Adam) after W warm-up iterations -> ms_per_step,
    iterations_per_s and the roofline block of the dominant kernel.  The timed iterations are a PINNED
    window of that fit -- iterations [STEP_I0, STEP_I0 + STEP_WINDOW) = [12, 32), whatever --steps and
    --warmup are: the fit runs its first 12 iterations, the engine's state is saved, the warm-up steps
    run and the state is put back; K > 20 timed steps replay the same 20 iterations (the state is put
    back after every 20: three elementwise copies of 3.8 MB each per 20 iterations, inside the timed
    region).  The scene shrinks while it is fitted (K = 250 k pairs at iteration 0, 190 k at 220), so a
    window that moved with --steps / --warmup timed another workload for every pair of flags;
  * the metric: an actual fit_video fit of one synthetic clip per rank (--clip-frames, default 60 =
    BASELINE configs[2]; README iteration counts 500 / 150 / 300, image-driven initialisation,
    densification, camera-only and joint stages, the reference's snapshots every 10th iteration) ->
    value = frames of all ranks / slowest rank's wall time.  Clips are independent: no data-path collective, "weak" scaling,
    one all-reduce(SUM) of a small metrics vector and one all-reduce(MAX) of the wall time.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

H, W, N_SPLATS = 480, 854, 60000
STEP_I0, STEP_WINDOW = 12, 20     # the timed step = iterations [12, 32) of the first-frame fit (see the docstring)
ITERS_PER_FRAME = (500 + 59 * (150 + 300)) / 60.0
TRAJ_NUM, TRAJ_OFFSET = 100, 2      # the README's --traj_num / --traj_offset: the clip fit renders its trajectories after every frame
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(kind, N, K, P):
    """Bytes one launch must move at minimum (DESIGN.md 'Kernels')."""
    if kind == "blend_bwd":
        return 44 * K + 24 * P + 40 * N        # gather splat records, read dL + aux, write reduced grads
    if kind == "blend_fwd":
        return 44 * K + 24 * P                 # gather splat records, write 4 planes + final_T + n_contrib
    if kind == "loss":
        return 48 * P                          # read render 16 + gt 16, write dL 16
    raise KeyError(kind)


STAGES = ["preprocess", "colscan", "scatter", "tile_sort", "blend_fwd", "loss", "blend_bwd", "pre_bwd_adam", "camera"]


def profile_read(lib):
    """Average milliseconds per stage from the HIP events the library recorded on the
    launch stream (gfl_profile_enable / gfl_profile_read, include/gflow_hip.h)."""
    import ctypes
    tot = (ctypes.c_double * len(STAGES))()
    cnt = (ctypes.c_int * len(STAGES))()
    lib.gfl_profile_read(tot, cnt, len(STAGES))
    return {STAGES[i]: tot[i] / cnt[i] for i in range(len(STAGES)) if cnt[i]}


def pmc_block(kind, window="first_frame"):
    """What the committed rocprofv3 counter passes of this same command say about a kernel of a pinned window
    (profiles/pmc_current.json, written by tools/profile_round.sh -> tools/summarise_profile.py; counters cannot be read
    from inside the process): ``traffic`` = HBM bytes per launch (FETCH_SIZE and WRITE_SIZE in separate passes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), and ``secondary`` = the issue bound SURVEY.md 8d
    asks for beside the HBM bound: VALU wave-instructions per launch, the share of the launch the VALUs were busy
    (SQ_ACTIVE_INST_VALU quad-cycles x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)) and the lane efficiency of a
    (splat, 8x8 block) unit.  ``window``: "first_frame" | "camera" | "joint" (bench.py's three pinned windows)."""
    path = os.path.join(ROOT, "profiles", "pmc_current.json")
    try:
        with open(path) as f:
            d = json.load(f)
        src = "profiles/pmc_current.json (" + d.get("tag", "?") + ", window " + window + ")"
        w = d["windows"][window] if "windows" in d else (d if window == "first_frame" else None)
        out = {"traffic": float(w["hbm_bytes_per_launch"][kind]), "traffic_source": src}
        v = w.get("valu", {}).get(kind)
        if v:
            out["secondary"] = dict(v, bound="valu", source=src)
        return out
    except (OSError, KeyError, ValueError, TypeError):
        return {"traffic": None}


def coresident_fits(dev, frames_n, snapshot_interval, levels=(1, 2, 3)):
    """Secondary measurement (``value`` stays the single-clip number the metric is quoted on): c = 1, 2, 3 clip fits AT
    THE SAME TIME on this GPU -- a stream per clip, the clips taking turns to enqueue 32 iterations each from one host
    thread, gflow_amd.fit_video.fit_clips_concurrent -- and the frames / s of all c clips together.  Clips are independent, so this is what a GPU does when there are more clips than
    GPUs; one fit alone leaves the chip partly idle (dependent launches, latency-bound kernels, blend tails)."""
    from gflow_amd import synthetic as S
    from gflow_amd import fit_video as FV
    clips = [FV.upload_clip(S.make_clip(frames_n, H, W, seed=100 + i, device=dev), dev) for i in range(max(levels))]
    cfg = dict(num_points=N_SPLATS, traj_num=TRAJ_NUM, traj_offset=TRAJ_OFFSET)
    FV.fit_clips_concurrent([c[:2] for c in clips], dev, cfg, snapshot_interval=snapshot_interval)      # warm-up
    torch.cuda.synchronize()
    out = {}
    for c in levels:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = FV.fit_clips_concurrent(clips[:c], dev, cfg, snapshot_interval=snapshot_interval)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        frames = sum(r["frames"] for r in res)
        out[str(c)] = {"clips": c, "frames_per_clip": frames_n, "wall_s": wall, "frames_per_s": frames / wall,
                       "iterations_per_s": sum(r["iterations"] for r in res) / wall,
                       "psnr_mean_db": sum(r["psnr_sum"] for r in res) / frames}
    # the same with the chip PARTITIONED between the clips (CU-masked streams, each clip on its share of every XCD)
    for c in levels:
        if c < 2:
            continue
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = FV.fit_clips_concurrent(clips[:c], dev, cfg, snapshot_interval=snapshot_interval, partition=True)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            frames = sum(r["frames"] for r in res)
            out[f"{c}_partitioned"] = {"clips": c, "frames_per_clip": frames_n, "wall_s": wall, "frames_per_s": frames / wall,
                                       "iterations_per_s": sum(r["iterations"] for r in res) / wall,
                                       "psnr_mean_db": sum(r["psnr_sum"] for r in res) / frames,
                                       "cus_per_clip": 8 * (32 // c)}
        except Exception as e:
            out[f"{c}_partitioned"] = {"error": f"{type(e).__name__}: {e}"}
    base = out[str(levels[0])]["frames_per_s"]
    for v in out.values():
        if "frames_per_s" in v:
            v["vs_one_clip"] = v["frames_per_s"] / base
    return out


def drop_in_levels(dev, Hh=None, Ww=None, Nn=None, repeats=3, n=20):
    """What the three levels of the drop-in cost per rasterisation forward + backward on one frame of the bench workload
     -- REPORTED, never asserted (tests/test_gpu_render_op.py used to hold a wall-clock ratio):
      five_operators_ms  the five msplat operators one by one, as /root/reference/gflow/utils/render.py:21-64 calls them
                         (the one-line ``import msplat`` swap; gflow_amd.msplat warns once when it sees that pattern),
      fused_render_ms    gflow_amd.render.render: ONE differentiable operator, two library calls,
      fit_iteration_ms   the fused fit iteration (the same rasterisation + losses + Adam): the level bench's value runs on.
    Best of ``repeats`` runs of ``n`` calls each (host time included: that is what a caller pays)."""
    import warnings
    import gflow_amd.render as R
    from gflow_amd import synthetic as S
    from gflow_amd.fused import FitEngine
    Hh, Ww, Nn = Hh or H, Ww or W, Nn or N_SPLATS
    frame = S.make_frame(Hh, Ww, seed=0)
    raw = S.init_splats(frame, Nn, seed=0, grown=True)
    act = [raw["xyz"], raw["scale"].abs(), torch.nn.functional.normalize(raw["rotate"]),
           torch.sigmoid(10 * raw["opacity"]), torch.sigmoid(raw["rgb"])]
    leaves = [v.to(dev).requires_grad_(True) for v in act]
    group = [*leaves, raw["intr"].to(dev), raw["extr"].to(dev), 0.0, Ww, Hh]
    g3 = ((torch.rand(3, Hh, Ww, device=dev) - 0.5) / (Hh * Ww)).contiguous()
    g1 = ((torch.rand(1, Hh, Ww, device=dev) - 0.5) / (Hh * Ww)).contiguous()
    images = {}

    def op_step(fused, keep=None):
        R.USE_FUSED = fused
        try:
            out = R.render_multiple(group, ["rgb", "uv", "depth", "depth_map"])
        finally:
            R.USE_FUSED = True
        torch.autograd.backward([out["rgb"], out["depth_map"]], [g3, g1])
        if keep is not None:
            images[keep] = out["rgb"].detach()

    eng = FitEngine(Ww, Hh, 2 * Nn, dev)
    eng.set_splats({k: raw[k] for k in ("xyz", "scale", "rotate", "opacity", "rgb")})
    eng.intr.copy_(raw["intr"].to(dev))
    eng.set_targets(frame["image"], frame["depth"])
    eng.hp.lr, eng.hp.lambda_depth, eng.hp.lambda_var = 0.0, 0.1, 10.0
    eng.reset_optimizer()

    def timed(fn):
        for _ in range(3):
            fn()
        best = float("inf")
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        return best * 1e3

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)              # (the slow level is run on purpose here)
        op_step(False, "ops")
        op_step(True, "fused")
        t_ops = timed(lambda: op_step(False))
    t_fused = timed(lambda: op_step(True))
    t_fit = timed(lambda: eng.iteration(use_graph=True))
    diff = float((images["ops"] - images["fused"]).abs().max())
    return {"five_operators_ms": t_ops, "fused_render_ms": t_fused, "fit_iteration_ms": t_fit,
            "five_operators_vs_fused_render": t_ops / t_fused, "fused_render_vs_fit_iteration": t_fused / t_fit,
            "rgb_max_abs_diff": diff, "size": f"{Hh}x{Ww}, {Nn} splats", "calls": f"best of {repeats} x {n}"}


# The clip's own stages, pinned like the first-frame window: frame CLIP_WINDOW_FRAME of the metric's clip, after a real fit
# of the frames before it -- N ~ 70 k splats with the piles of thirty frames' densification, frozen colours, flow and still
# terms -- its camera-only stage at iterations [CAM_I0, +20) (backward blend <6>, forward <3> with the footprint workgroups,
# fused_camera_adam) and its joint stage at [JOINT_I0, +20) (backward blend <7>: behind both densifications, 0 and 99).
# 98 % of the metric's 27 050 iterations are of these two kinds (59 x 150 and 59 x 300); the first-frame window describes 500.
CLIP_WINDOW_FRAME, CAM_I0, JOINT_I0 = 30, 60, 120
STAGE_SHARE = {"first_frame": 500 / 27050.0, "camera": 59 * 150 / 27050.0, "joint": 59 * 300 / 27050.0}


def measure_window(eng, stepper, saved, args, barrier, lib, units=True):
    """Time the pinned window the stepper stands at the start of: W warm-up steps, then EXACTLY K steps between two barriers
    (-> elapsed: what ms_per_step is quoted on), then the same K steps ``args.repeats`` times more, each between barriers of
    its own (-> repeats: a 4 ms region on a box whose minutes differ by 5 % says little alone), then once more with the
    library's HIP events around every stage (-> stage_ms) and the pair count of every iteration read (-> K).  K > 20 timed
    steps replay the same 20 iterations: the state is put back after every 20."""
    import gc

    def run_window(n):
        done = 0
        while done < n:
            k = min(STEP_WINDOW, n - done)
            if done:
                eng.restore_state(saved)
            stepper.run(k)
            done += k

    # (at least one whole window, a restore and 4 + 2 + 1 steps untimed: the replayed graphs hold one, two or four iterations,
    #  and the one that follows a restore starts on the exact binning path -- every variant is captured before the clock runs)
    # The interpreter's collection comes BEFORE the warm-up steps, not between them and the clock: it takes tens of
    # milliseconds, the device sat idle meanwhile, and the first launches of a 4 ms timed region then ran on clocks that had
    # dropped (--steps 20: 0.208 ms per step against 0.196 at --steps 200, the same kernels).
    gc.collect()
    gc.disable()          # (a generation-2 collection of the interpreter inside a 4 ms timed region is not the kernels' time)
    try:
        run_window(max(args.warmup, STEP_WINDOW + 7))
        eng.restore_state(saved)
        # timed region: exactly K steps, no instrumentation (an event pair between two kernels
        # opens a 5-10 us bubble on the stream, measured with rocprofv3)
        barrier()
        void0 = int(eng.overflow[1].item())
        t0 = time.perf_counter()
        run_window(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        # iterations of the timed region that stepped nothing because a tile outgrew its reserved region (FitEngine.settle_overflow)
        void_iterations = int(eng.overflow[1].item()) - void0
        reps = []
        for _ in range(max(0, args.repeats)):
            eng.restore_state(saved)
            barrier()
            t0 = time.perf_counter()
            run_window(args.steps)
            barrier()
            reps.append((time.perf_counter() - t0) / args.steps * 1e3)
    finally:
        gc.enable()
    # the same K steps again with HIP events recorded by the library on the launch stream
    # around every stage: per-kernel durations for the roofline block; and the window's pair counts K
    stage_ms = {}
    n_k = min(args.steps, STEP_WINDOW)
    k_dev = torch.zeros(STEP_WINDOW, dtype=torch.int32, device=eng.dev)
    if not args.no_stage_pass:
        from gflow_amd.fused import set_profile
        set_profile((1 << len(STAGES)) - 1)
        for i in range(args.steps):
            if i % STEP_WINDOW == 0:
                eng.restore_state(saved)
            stepper()
            k_dev[i % STEP_WINDOW:i % STEP_WINDOW + 1].copy_(eng.tile_offsets[eng.T:eng.T + 1])
        torch.cuda.synchronize()
        set_profile(0)
        stage_ms = profile_read(lib)
    else:
        eng.restore_state(saved)
        for i in range(n_k):
            stepper()
            k_dev[i:i + 1].copy_(eng.tile_offsets[eng.T:eng.T + 1])
    eng.check_overflow()                   # (the stepper is driven directly here: no train() looks at the pair lists' flag)
    ks = k_dev[:n_k].cpu().tolist()
    out = {"elapsed": elapsed, "ms_per_step": elapsed / args.steps * 1e3, "timed_region_s": elapsed, "stage_ms": stage_ms,
           "K_list": ks, "K_mean": sum(ks) / len(ks),          # mean over the window's iterations: what the kernels' average durations belong to
           "void_iterations": void_iterations, "reserved_on": bool(eng._reserved_flag()), "splats": int(eng.N)}
    if reps:
        r = sorted(reps)
        out["repeats"] = {"n": len(r), "median": r[len(r) // 2], "min": r[0], "max": r[-1], "steps_each": args.steps,
                          "note": "the same window timed again in this process, a barrier + synchronise on both sides of "
                                  "each; ms per step"}
    if units:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from unit_stats import unit_stats
            out["work"] = unit_stats(eng)          # of the window's LAST iteration's forward
        except Exception as e:
            out["work"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def window_report(m, name, kernels, N, P):
    """A window's entry of the JSON line: step time, its spread, the work, the blend kernels against the HBM roofline."""
    K = m["K_mean"]
    kern = {}
    for kind in ("blend_fwd", "loss", "blend_bwd"):
        ms = m["stage_ms"].get(kind)
        if ms:
            b = algorithmic_bytes(kind, N, K, P)
            kern[kind] = {"kernel": kernels.get(kind), "ms": ms, "algorithmic_bytes": b, "GBps": b / (ms * 1e-3) / 1e9,
                          "frac_of_hbm_peak": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    out = {"name": name, "share_of_the_clips_iterations": STAGE_SHARE.get(name), "ms_per_step": m["ms_per_step"],
           "timed_region_s": m["timed_region_s"], "ms_per_step_repeats": m.get("repeats"), "splats": m["splats"],
           "K_mean": K, "K_first": m["K_list"][0], "K_last": m["K_list"][-1], "void_iterations": m["void_iterations"],
           "stage_ms": m["stage_ms"], "stage_sum_ms": sum(m["stage_ms"].values()) if m["stage_ms"] else None,
           "kernels": kern, "work": m.get("work"),
           "algorithmic_bytes_per_iteration": 724 * N + 124 * K + 96 * P}
    if m["stage_ms"]:
        out["end_to_end_algorithmic_GBps"] = out["algorithmic_bytes_per_iteration"] / (m["ms_per_step"] * 1e-3) / 1e9
    return out


def measure_clip_windows(dev, rank, args, barrier, lib, which=("camera", "joint")):
    """The two mid-clip windows (see CLIP_WINDOW_FRAME above).  The frames before the window's frame are really fitted
    (README recipe, snapshots on), then the frame's camera-only stage is set up and timed, run for real, and the joint stage
    set up and timed: the trainer's own calls (fit_video.begin_frame / stage_kwargs), no shortcut."""
    from gflow_amd import synthetic as S
    from gflow_amd import fit_video as FV
    f_i = min(CLIP_WINDOW_FRAME, args.clip_frames - 1)
    if f_i < 1:
        return None
    c = dict(FV.DEFAULTS)
    c.update(num_points=N_SPLATS)
    frames = FV.upload_clip(S.make_clip(f_i + 1, H, W, seed=rank, device=dev), dev)      # (a clip's frames do not depend on its length)
    keep = {}
    FV.fit_clip(frames[:f_i], dev, dict(num_points=N_SPLATS), seed=rank, snapshot_interval=args.snapshot_interval, keep=keep)
    tr = keep["trainer"]
    common = dict(lambda_rgb=c["lambda_rgb"], lambda_depth=c["lambda_depth"], lambda_scale=c["lambda_scale"],
                  densify_occ_percent=c["densify_occ_percent"], densify_err_thre=c["densify_err_thre"],
                  densify_err_percent=c["densify_err_percent"])
    out = {"frame": f_i, "note": f"frame {f_i} of the metric's clip after a real fit of frames [0, {f_i}); camera-only stage "
                                 f"iterations [{CAM_I0}, {CAM_I0 + STEP_WINDOW}), joint stage [{JOINT_I0}, {JOINT_I0 + STEP_WINDOW})"}
    P = H * W
    FV.begin_frame(tr, frames, f_i)
    # camera-only stage: a stepper of our own for the window, then the pose is put back and the stage is run for real
    pose0 = tr.pose.detach().clone()
    st = tr.make_stepper(**FV.stage_kwargs(c, frames, f_i, "camera"), snapshot_interval=0, **common)
    eng = tr.engine
    if "camera" in which:
        st.run(CAM_I0)
        m = measure_window(eng, st, eng.save_state(), args, barrier, lib)
        out["camera"] = window_report(m, "camera", {"blend_fwd": "fused_blend_fwd_kernel<3>", "blend_bwd": "fused_blend_bwd_kernel<6>"},
                                      m["splats"], P)
    del st
    eng.pose.copy_(pose0)
    if "joint" not in which:
        return out
    tr.train(**FV.stage_kwargs(c, frames, f_i, "camera"), snapshot_interval=args.snapshot_interval, lazy_images=True, **common)
    # joint stage (the stepper's set-up carries the moving splats along the flow, appends the occlusion splats at iteration 0
    # and the error splats at 99: the window lies behind both)
    st = tr.make_stepper(**FV.stage_kwargs(c, frames, f_i, "joint"), snapshot_interval=0, **common)
    eng = tr.engine
    st.run(JOINT_I0)
    m = measure_window(eng, st, eng.save_state(), args, barrier, lib)
    out["joint"] = window_report(m, "joint", {"blend_fwd": "fused_blend_fwd_kernel<0>", "blend_bwd": "fused_blend_bwd_kernel<7>"},
                                 m["splats"], P)
    del st, tr, keep
    return out


def cpu_baseline():
    """The oracle's fit iteration on the host cores: 3 warm-ups + 20 timed iterations of
    the bench workload (480x854, 60 000 splats) and of the C1-size workload (10 000 splats)."""
    from gflow_amd import synthetic as S
    from oracle.fit_oracle import OracleFit
    # eager torch on hundreds of threads thrashes on the many small index ops of the
    # oracle (measured: 256 threads are ~100x slower than 8); use at most 16
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    frame = S.make_frame(H, W, seed=0)
    out = {}
    for name, n_splats in (("c2", N_SPLATS), ("c1", 10000)):
        raw = S.init_splats(frame, n_splats, seed=0, grown=True)
        fit = OracleFit(raw, raw["intr"], frame, lr=4e-3, iterations=500, lambda_depth=0.1, lambda_var=10.0)
        for _ in range(3):
            fit.step()                               # warm-up (allocator, thread pool)
        timers = {}
        t0 = time.time()
        n = 0
        while n < 20 and (n < 5 or time.time() - t0 < 40.0):
            fit.step(timers)
            n += 1
        dt = (time.time() - t0) / n
        out[name] = {"splats": n_splats, "iterations_timed": n, "ms_per_step": dt * 1000.0,
                     "phase_ms": {k: v / n * 1000.0 for k, v in timers.items()}}
    dt = out["c2"]["ms_per_step"] / 1000.0
    return {"value": (1.0 / dt) / ITERS_PER_FRAME, "unit": "frames/s", "cores": cores, "kind": "port",
            "host_cores": os.cpu_count(),
            "sample": f"{cores} of {os.cpu_count()} host threads (eager PyTorch gets slower beyond 16 on this workload); "
                      f"{out['c2']['iterations_timed']} fit iterations (after 3 warm-ups) of the same 480x854 / 60k-splat "
                      f"frame with the eager-PyTorch CPU oracle (own restatement; the reference has no CPU rasteriser), "
                      f"{dt * 1000:.0f} ms/iteration; frames/s = iterations/s / {ITERS_PER_FRAME:.2f} iterations per frame; "
                      f"C1-size (10k splats) sample alongside",
            "ms_per_step": out["c2"]["ms_per_step"], "phase_ms": out["c2"]["phase_ms"], "c1_10k_splats": out["c1"]}


def reduce_and_report(local, dist, red_dev, rank, world, args, backend, size=None):
    """The two metric all-reduces (SUM of a small vector, MAX of the wall times) and the JSON line rank 0 prints.
    ``local``: this rank's measurements -- elapsed (s of the timed steps), steps, psnr_step, K (splat-tile pairs of
    ITS scene), clip (fit_clip's metrics dict or None), clip_wall, kernels_ms, stage_ms.  Returns the dict on rank 0,
    None elsewhere.  (A function of its own so that tests/test_host_logic.py can run it under 2-rank gloo.)"""
    Hh, Ww, Nn = size or (H, W, N_SPLATS)
    clip = local["clip"]
    vec = [local["elapsed"], float(local["steps"]), local["psnr_step"], float(local["K"])]
    if clip is not None:
        vec += [clip[k] for k in ("frames", "iterations", "rasterisations", "psnr_sum", "splats_final")] + [clip.get("clips", 1)]
        # every rank's OWN clip wall time in a slot of its own: the one SUM hands rank 0 all of them (the shard's imbalance)
        own = [0.0] * world
        own[rank] = float(local.get("clip_wall_own", local["clip_wall"]))
        vec += own
    stats = torch.tensor(vec, dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([local["elapsed"], local["clip_wall"]], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    if rank != 0:
        return None
    wall = float(tmax[0].item())
    it_per_s = float(stats[1].item()) / wall          # whole job: iterations of all ranks / slowest rank's time
    K_mean = float(stats[3].item()) / world           # every rank has its own scene: the mean over the ranks
    P = Hh * Ww
    roof = {}
    for kind, ms in local["kernels_ms"].items():
        if ms:
            # rank 0's kernels on rank 0's scene
            b = algorithmic_bytes(kind, Nn, local["K"], P)
            roof[kind] = {"ms": ms, "algorithmic_bytes": b, "GBps": b / (ms * 1e-3) / 1e9}
    roofline = roofline_first = None
    if roof:
        dom = max(roof, key=lambda k: roof[k]["ms"])
        roofline_first = {"bound": "hbm", "kernel": dom, "window": "first_frame", "achieved": roof[dom]["GBps"],
                          "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": roof[dom]["GBps"] / HBM_PEAK_GBPS}
        roofline_first.update(pmc_block(dom, "first_frame"))
        roofline = roofline_first
    # The dominant kernel of the METRIC's workload: 65 % of a clip's iterations are joint-stage iterations of later frames
    # (backward blend <7>, frozen colours), 33 % camera-only (<6>), 2 % first-frame (<10>) -- so the roofline block describes
    # the joint-stage window when it was measured, and the first-frame window's block stays beside it.
    cwin = local.get("clip_windows") or {}
    jw = cwin.get("joint") if isinstance(cwin, dict) else None
    if jw and jw.get("kernels"):
        kk = jw["kernels"]
        dom = max(kk, key=lambda k: kk[k]["ms"])
        roofline = {"bound": "hbm", "kernel": kk[dom]["kernel"] or dom, "window": "joint (step_window_clip)",
                    "achieved": kk[dom]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": kk[dom]["GBps"] / HBM_PEAK_GBPS,
                    "algorithmic_bytes_per_launch": kk[dom]["algorithmic_bytes"], "avg_launch_ms": kk[dom]["ms"],
                    "share_of_the_clips_iterations": STAGE_SHARE["joint"]}
        roofline.update(pmc_block(dom, "joint"))
    derived = it_per_s / ITERS_PER_FRAME
    workload = ("configs[1]" if (Hh, Ww, Nn) == (480, 854, 60000) else "other") + f": {Hh}x{Ww}, {Nn} splats"
    if clip is not None:
        cw = float(tmax[1].item())
        frames_all, iters_all = float(stats[4].item()), float(stats[5].item())
        value = frames_all / cw
        clip_out = {"frames_per_rank": args.clip_frames, "wall_s": cw, "iterations": iters_all,
                    "iterations_per_s": iters_all / cw, "rasterisations_per_s": float(stats[6].item()) / cw,
                    "psnr_mean_db": float(stats[7].item()) / frames_all,
                    "splats_final_mean": float(stats[8].item()) / max(float(stats[9].item()), 1.0),
                    "snapshot_interval": args.snapshot_interval,
                    "clips_per_rank": int(round(float(stats[9].item()) / world)),
                    "rank_wall_s": [float(v) for v in stats[10:10 + world].tolist()],
                    # rank 0's clips: iterations that stepped nothing because a tile outgrew its reserved region, and were
                    # made up for (they are inside wall_s; `iterations` counts the ones that stepped)
                    "void_iterations": (clip or {}).get("void_iterations")}
        workload += (f"; value = measured fit_video fit of one {args.clip_frames}-frame rigid synthetic clip per GPU "
                     f"(configs[2]: iterations 500 first / 150 camera-only + 300 joint per later frame, densification "
                     f"on, snapshots every {args.snapshot_interval} iterations, --traj_num {TRAJ_NUM} --traj_offset {TRAJ_OFFSET}: "
                     f"trainer.eval + project_points after every frame, /root/reference/README.md:114-115); ms_per_step = iterations "
                     f"[{STEP_I0}, {STEP_I0 + STEP_WINDOW}) of a first-frame fit on the grown scene (lambda rgb/depth/var = "
                     f"1/0.1/10), the same window for every --steps / --warmup")
    else:
        value, clip_out = derived, None
        workload += "; value DERIVED from the first-frame fit iteration (no clip fit in this run)"
    bytes_per_iteration = 724 * Nn + 124 * K_mean + 96 * P
    return {
        "metric": "GFlow fit_video frames/sec (fwd+bwd+step) @60k Gaussians 480p",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1000.0,
        "timed_region_s": wall,                      # the K timed steps of the first-frame window: max over ranks
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "iterations_per_frame": ITERS_PER_FRAME, "splat_tile_pairs_K": K_mean,
                   "step_window": {"iterations": [STEP_I0, STEP_I0 + STEP_WINDOW], "K_first": local.get("K_first"),
                                   "K_last": local.get("K_last"), "K_mean": local["K"],
                                   "timed_region_s": local["elapsed"], "splats": Nn,
                                   "work": (local.get("window") or {}).get("work"),
                                   # the window's first iteration follows restore_state and bins on the exact path (three
                                   # launches); the other 19 bin into the tile regions the iteration before reserved (one)
                                   "binning": ("reserved tile regions in %d of %d iterations" % (STEP_WINDOW - 1, STEP_WINDOW))
                                   if local.get("reserved_on") else "exact path",
                                   "void_iterations": local.get("void_iterations")},
                   # (--clips-per-gpu c > 1: the throughput mode of a node with more clips than GPUs -- every rank fits c clips
                   #  at the same time; the metric is quoted on 1)
                   "parallelism": f"clip-sharded x{world}", "clips_per_gpu": max(1, getattr(args, "clips_per_gpu", 1)),
                   "collective_backend": backend},
        "value_kind": "measured clip fit" if clip is not None else "derived from the step",
        "clip_fit": clip_out,
        "iterations_per_s": it_per_s,
        "rasterisations_fwd_bwd_per_s": it_per_s,
        "frames_per_s_derived_from_step": derived,
        "psnr_step_mean_db": float(stats[2].item()) / world,
        "roofline": roofline,
        "roofline_first_frame_window": roofline_first,
        "kernels": roof,
        "stage_ms": local["stage_ms"],
        # the same window timed `--repeats` times more in this process (a 4 ms region alone cannot carry a 3-5 % claim)
        "ms_per_step_repeats": (local.get("window") or {}).get("repeats"),
        "step_window_camera": cwin.get("camera") if isinstance(cwin, dict) else None,
        "step_window_clip": jw,
        "clip_windows": {k: v for k, v in cwin.items() if k not in ("camera", "joint")} if isinstance(cwin, dict) else cwin,
        "clip_iteration_model": clip_iteration_model(local, cwin, clip_out, world),
        # whole job (all ranks) and per GPU
        "end_to_end_algorithmic_GBps": bytes_per_iteration * it_per_s / 1e9,
        "end_to_end_algorithmic_GBps_per_gpu": bytes_per_iteration * it_per_s / world / 1e9,
    }


def clip_iteration_model(local, cw, clip_out, world=1):
    """Do the three pinned windows explain the clip fit?  Their step times weighted by the stages' shares of the clip's
    iterations, against the clip fit's own wall time per iteration (which also holds densification events, snapshots,
    frame boundaries and host time)."""
    try:
        w1 = local["elapsed"] / local["steps"] * 1e3
        cam, joint = cw["camera"]["ms_per_step"], cw["joint"]["ms_per_step"]
        model = STAGE_SHARE["first_frame"] * w1 + STAGE_SHARE["camera"] * cam + STAGE_SHARE["joint"] * joint
        out = {"windows_weighted_ms_per_iteration": model,
               "weights": STAGE_SHARE, "note": "first-frame, camera-only and joint window step times weighted by the stages' "
                                               "shares of a 60-frame clip's 27 050 iterations (mid-clip state: frame %d)" % CLIP_WINDOW_FRAME}
        if clip_out:
            # (iterations of all ranks' clips; a rank's clips run one after another unless --clips-per-gpu says otherwise)
            meas = clip_out["wall_s"] / max(clip_out["iterations"] / world, 1) * 1e3
            out["clip_fit_ms_per_iteration"] = meas
            out["unexplained_frac"] = (meas - model) / meas
        return out
    except (KeyError, TypeError, ZeroDivisionError):
        return None


def respawn(args):
    """--gpus N without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-pass", action="store_true",
                    help="skip the second, event-instrumented pass (used under rocprofv3)")
    ap.add_argument("--size", default=None,
                    help="HxWxN to time another configuration (e.g. 720x1280x200000 = BASELINE configs[4]); "
                         "the default and the metric are configs[1]")
    ap.add_argument("--no-clip", action="store_true",
                    help="skip the clip fit; value is then DERIVED from the step (iterations/s / 450.83) and says so")
    ap.add_argument("--clip-frames", type=int, default=60,
                    help="frames of the clip whose fit is the metric (BASELINE configs[2]: ~60)")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="clips every rank fits AT THE SAME TIME on its GPU (fit_video.fit_clips_concurrent; throughput "
                         "mode of a node with more clips than GPUs).  The metric is quoted on 1")
    ap.add_argument("--no-coresident", action="store_true",
                    help="skip the secondary table of 1 / 2 / 3 clip fits sharing this GPU (clips_per_gpu)")
    ap.add_argument("--coresident-frames", type=int, default=8, help="frames per clip of that secondary table")
    ap.add_argument("--no-clip-windows", action="store_true",
                    help="skip the two mid-clip step windows (camera-only and joint stage of frame %d)" % CLIP_WINDOW_FRAME)
    ap.add_argument("--only-window", choices=("first_frame", "camera", "joint"), default=None,
                    help="profiling runs (tools/profile_round.sh): nothing but this pinned window -- no clip fit, no secondary "
                         "tables -- so that the LAST launches of every kernel in a rocprofv3 trace are the window's")
    ap.add_argument("--repeats", type=int, default=10,
                    help="time every pinned window this many times more (ms_per_step_repeats: median / min / max)")
    ap.add_argument("--no-drop-in-levels", action="store_true",
                    help="skip the secondary table of what the three levels of the drop-in cost (five operators / fused "
                         "render operator / fused fit iteration)")
    ap.add_argument("--collective", action="store_true",
                    help="initialise the process group and run the barriers and the two metric all-reduces even when the "
                         "world is ONE rank (under torchrun): the RCCL path of an N > 1 run on a one-GPU box")
    ap.add_argument("--snapshot-interval", type=int, default=10,
                    help="snapshots of the clip fit (the reference keeps three images every 10th iteration, "
                         "trainer.py:573-582); 0 = none")
    args = ap.parse_args()

    if args.only_window:
        args.no_clip = args.no_coresident = args.no_drop_in_levels = args.no_cpu_baseline = True
    global H, W, N_SPLATS
    if args.size:
        H, W, N_SPLATS = (int(v) for v in args.size.lower().split("x"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.exit("bench.py needs a HIP device")
    # (ranks share devices only when a box has fewer GPUs than ranks -- a functional run, e.g. CI; RCCL
    # refuses two ranks on one device, so the two tiny metric all-reduces then go over gloo)
    shared = world > n_dev
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    dist = None
    backend = None
    if world > 1 or (args.collective and "WORLD_SIZE" in os.environ):
        import torch.distributed as dist
        backend = "gloo" if shared else "nccl"
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    red_dev = torch.device("cpu") if backend == "gloo" else dev

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    from gflow_amd import _lib
    from gflow_amd import synthetic as S
    from gflow_amd import fit_video as FV
    from gflow_amd.trainer import SimpleGaussian
    lib = _lib.load()

    # -------------------------------------------------------------- the metric
    # (first: its seconds of GPU work also bring the device to its working clocks before the short step timing below)
    clip = None
    clip_wall = own_wall = 0.0
    if not args.no_clip:
        c = max(1, args.clips_per_gpu)
        clips = [FV.upload_clip(S.make_clip(args.clip_frames, H, W, seed=rank * c + j, device=dev), dev) for j in range(c)]
        cfg = dict(num_points=N_SPLATS, traj_num=TRAJ_NUM, traj_offset=TRAJ_OFFSET)
        FV.fit_clip(clips[0][:2], dev, cfg, seed=rank, snapshot_interval=args.snapshot_interval)      # warm-up
        barrier()
        t0 = time.perf_counter()
        if c == 1:
            clip = FV.fit_clip(clips[0], dev, cfg, seed=rank, snapshot_interval=args.snapshot_interval)
        else:
            res = FV.fit_clips_concurrent(clips, dev, cfg, seeds=[rank * c + j for j in range(c)],
                                          snapshot_interval=args.snapshot_interval)
            clip = {k: sum(r[k] for r in res) for k in FV.NUMERIC_KEYS}
        torch.cuda.synchronize()
        own_wall = time.perf_counter() - t0
        barrier()
        clip_wall = time.perf_counter() - t0
        del clips

    # ---------------------------------------------------------------- the step
    # (on a stream of its own, like the clip fit: the default stream is HIP's legacy NULL stream, which synchronises with
    #  every other blocking stream of the process -- after the clip fit has created its fit / copy / capture streams the
    #  same 200 graph replays took 0.26 ms per step there instead of 0.20)
    step_stream = torch.cuda.Stream(device=dev)
    step_stream.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(step_stream)
    frame = S.make_frame(H, W, seed=rank)
    raw = S.init_splats(frame, N_SPLATS, seed=rank, grown=True)
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=N_SPLATS, device=dev, seed=rank)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
        tr._attributes[k] = raw[k].to(dev)
    kw = dict(lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, move_mask=frame["move_mask"],
              densify_interval=0, snapshot_interval=0)
    stepper = tr.make_stepper(iterations=500, **kw)
    eng = tr.engine
    # the pinned window: iterations [STEP_I0, STEP_I0 + STEP_WINDOW) of this fit, whatever --steps / --warmup are
    stepper.run(STEP_I0)
    m1 = measure_window(eng, stepper, eng.save_state(), args, barrier, lib)
    elapsed, kern_all, K, ks = m1["elapsed"], m1["stage_ms"], m1["K_mean"], m1["K_list"]
    void_iterations, reserved_on = m1["void_iterations"], m1["reserved_on"]
    kern = {k: kern_all[k] for k in ("blend_fwd", "loss", "blend_bwd") if k in kern_all}
    psnr_step = float(tr.psnr_of(stepper.last_render))
    del stepper, tr, eng

    # ------------------------------------------------ the clip's own two stages, pinned the same way (rank 0's scene)
    clip_windows = None
    want_cw = (args.only_window in ("camera", "joint")) or (not args.only_window and not args.no_clip_windows and not args.no_clip)
    # (rank 0 only, and with a LOCAL synchronisation in place of the collective barrier: a secondary measurement must not be able
    #  to leave the other ranks of an N > 1 run waiting in a barrier it never reaches)
    if want_cw and rank == 0 and (H, W, N_SPLATS) == (480, 854, 60000):
        try:
            clip_windows = measure_clip_windows(dev, rank, args, torch.cuda.synchronize, lib,
                                                which=(args.only_window,) if args.only_window else ("camera", "joint"))
        except Exception as e:                       # secondary windows must not cost the line
            clip_windows = {"error": f"{type(e).__name__}: {e}"}

    local = {"elapsed": elapsed, "steps": args.steps, "psnr_step": psnr_step, "K": K, "K_first": ks[0], "K_last": ks[-1],
             "clip": clip, "clip_wall": clip_wall, "clip_wall_own": own_wall, "kernels_ms": kern, "stage_ms": kern_all,
             "void_iterations": void_iterations, "reserved_on": reserved_on, "window": m1, "clip_windows": clip_windows}
    out = reduce_and_report(local, dist, red_dev, rank, world, args, backend)
    if out is not None:
        if world == 1 and not args.no_clip and not args.no_coresident:
            try:
                out["clips_per_gpu"] = coresident_fits(dev, args.coresident_frames, args.snapshot_interval)
            except Exception as e:                   # a secondary table must not cost the line
                out["clips_per_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_drop_in_levels:
            try:
                out["drop_in_levels"] = drop_in_levels(dev)
            except Exception as e:                   # a secondary table must not cost the line
                out["drop_in_levels"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
