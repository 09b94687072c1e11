"""Benchmark of the hot path: one "step" = one optimisation iteration of the GFlow
first-frame fit (rasterise forward, photometric+SSIM+depth+var loss, rasterise
backward, Adam) at BASELINE.json configs[1]: 480x854, 60 000 splats, synthetic frame.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Each rank fits its own synthetic clip frame (no data-path collective, "weak" scaling);
one all-reduce(MAX) of the wall time and one all-reduce(SUM) of a small metrics vector
at the end.  Rank 0 prints ONE JSON line.

value = frames/s: iterations/s divided by the iterations GFlow spends per frame with
the README flags on a 60-frame clip, (500 + 59*(150+300))/60 = 450.83 (README.md:89-107).
Inputs are resident in HBM before the timed region.
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
ITERS_PER_FRAME = (500 + 59 * (150 + 300)) / 60.0
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


def pmc_traffic(kind):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same command (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950); counters cannot be read from inside the process,
    so the number comes from profiles/pmc_current.json (written by tools/summarise_profile.py)."""
    path = os.path.join(ROOT, "profiles", "pmc_current.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return {"traffic": float(d["hbm_bytes_per_launch"][kind]),
                "traffic_source": "profiles/pmc_current.json (" + d.get("tag", "?") + ")"}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def clip_fit(dev, frames_n=4):
    """End-to-end check of the derived frames/s: an actual fit of a short synthetic clip through
    gflow_amd.fit_video.fit_clip (image-driven initialisation, densification, camera-only and
    joint stages, all host work included; clip synthesis excluded).  Real fits are heavier per
    iteration than the steady-state step timed above: densification piles splats into few tiles."""
    from gflow_amd import synthetic as S
    from gflow_amd import fit_video as FV
    frames = S.make_clip(frames_n, H, W, seed=0)
    FV.fit_clip(frames[:2], dev, dict(num_points=N_SPLATS), seed=0)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = FV.fit_clip(frames, dev, dict(num_points=N_SPLATS), seed=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # frame 0 costs 500 iterations, every later frame 150 + 300: extrapolate to the 60-frame clip
    it_per_s = m["iterations"] / dt
    return {"frames": frames_n, "wall_s": dt, "iterations": m["iterations"], "iterations_per_s": it_per_s,
            "frames_per_s_this_clip": frames_n / dt, "frames_per_s_60_frame_clip": it_per_s / ITERS_PER_FRAME,
            "psnr_mean_db": m["psnr_sum"] / frames_n, "splats_final": m["splats_final"]}


def cpu_baseline(seconds_budget=25.0):
    """The oracle's fit iteration on the host cores, same workload, bounded sample."""
    from gflow_amd import synthetic as S
    from oracle.fit_oracle import OracleFit
    # eager torch on hundreds of threads thrashes on the many small index ops of the
    # oracle (measured: 256 threads are ~100x slower than 8); use at most 16
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    frame = S.make_frame(H, W, seed=0)
    raw = S.init_splats(frame, N_SPLATS, seed=0, grown=True)
    fit = OracleFit(raw, raw["intr"], frame, lr=4e-3, iterations=500, lambda_depth=0.1, lambda_var=10.0)
    fit.step()                                   # warm-up (allocator, thread pool)
    t0 = time.time()
    n = 0
    timers = {}
    while n < 1 or (time.time() - t0 < seconds_budget and n < 8):
        fit.step(timers)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": (1.0 / dt) / ITERS_PER_FRAME, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} fit iterations (after 1 warm-up) of the same 480x854 / 60k-splat frame with the "
                      f"eager-PyTorch CPU oracle (own restatement; the reference has no CPU rasteriser), "
                      f"{dt * 1000:.0f} ms/iteration",
            "ms_per_step": dt * 1000.0,
            "phase_ms": {k: v / n * 1000.0 for k, v in timers.items()}}


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
                    help="skip the end-to-end fit of a short synthetic clip (extra field clip_fit)")
    args = ap.parse_args()

    global H, W, N_SPLATS
    if args.size:
        H, W, N_SPLATS = (int(v) for v in args.size.lower().split("x"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    from gflow_amd import _lib
    from gflow_amd import synthetic as S
    from gflow_amd.trainer import SimpleGaussian
    lib = _lib.load()

    frame = S.make_frame(H, W, seed=rank)
    raw = S.init_splats(frame, N_SPLATS, seed=rank, grown=True)
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=N_SPLATS, device=dev, seed=rank)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
        tr._attributes[k] = raw[k].to(dev)
    total = args.warmup + args.steps
    kw = dict(lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, move_mask=frame["move_mask"],
              densify_interval=0, snapshot_interval=0)
    stepper = tr.make_stepper(iterations=500, **kw)
    for _ in range(args.warmup):
        stepper()
    # ---- timed region: exactly K steps, no instrumentation (an event pair between two kernels
    # opens a 5-10 us bubble on the stream, measured with rocprofv3)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stepper()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # ---- the same K steps again with HIP events recorded by the library on the launch stream
    # around every stage: per-kernel durations for the roofline block
    kern_all = {}
    if not args.no_stage_pass:
        from gflow_amd.fused import set_profile
        set_profile((1 << len(STAGES)) - 1)
        for _ in range(args.steps):
            stepper()
        torch.cuda.synchronize()
        set_profile(0)
        kern_all = profile_read(lib)
    kern = {k: kern_all[k] for k in ("blend_fwd", "loss", "blend_bwd") if k in kern_all}

    K = tr.engine.K if tr.engine is not None else int(tr.last_K)
    psnr = float(tr.psnr_of(stepper.last_render))
    stats = torch.tensor([elapsed, float(args.steps), psnr, float(K)], dtype=torch.float64, device=dev)
    tmax = stats[:1].clone()
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    wall = float(tmax.item())
    total_steps = float(stats[1].item())
    if rank == 0:
        it_per_s = total_steps / wall
        P = H * W
        roof = {}
        for kind, ms in kern.items():
            if ms:
                b = algorithmic_bytes(kind, N_SPLATS, K, P)
                roof[kind] = {"ms": ms, "algorithmic_bytes": b, "GBps": b / (ms * 1e-3) / 1e9}
        roofline = None
        if roof:
            dom = max(roof, key=lambda k: roof[k]["ms"])
            roofline = {"bound": "hbm", "kernel": dom, "achieved": roof[dom]["GBps"], "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": roof[dom]["GBps"] / HBM_PEAK_GBPS}
            roofline.update(pmc_traffic(dom))
        out = {
            "metric": "GFlow fit_video frames/sec (fwd+bwd+step) @60k Gaussians 480p",
            "value": it_per_s / ITERS_PER_FRAME,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1000.0,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("configs[1]" if (H, W, N_SPLATS) == (480, 854, 60000) else "other") +
                                   f": first-frame fit iteration, {H}x{W}, {N_SPLATS} splats (grown footprint), "
                                   "lambda rgb/depth/var = 1/0.1/10, one clip per GPU",
                       "iterations_per_frame": ITERS_PER_FRAME, "splat_tile_pairs_K": K,
                       "parallelism": f"clip-sharded x{world}"},
            "iterations_per_s": it_per_s,
            "rasterisations_fwd_bwd_per_s": it_per_s,
            "psnr_mean_db": float(stats[2].item()) / world,
            "roofline": roofline,
            "kernels": roof,
            "stage_ms": kern_all,
            "end_to_end_algorithmic_GBps": (724 * N_SPLATS + 124 * K + 96 * P) * it_per_s / world / 1e9,
        }
        if world == 1 and not args.no_clip:
            out["clip_fit"] = clip_fit(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
