"""Where a snapshot's cost goes, from the kernel trace of a clip fit (tools/quick_clip_trace.sh <tag> leaves
gpurun_out/qc_<tag>/r_kernel_trace.csv):  python tools/snapshot_gaps.py gpurun_out/qc_<tag>/r_kernel_trace.csv
The fit's own kernels form a chain on one queue; an ITERATION starts at a preprocess kernel.  For every iteration: its span
(start of its preprocess to start of the next one's), the sum of its kernels' durations, the idle time between them; grouped by
what the iteration is -- plain on reserved regions, plain on the exact path (the one in front of a looked-at iteration),
looked-at (exact path + a snapshot staged behind it)."""
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.split("(")[0].replace("void ", "").replace("gfl::", "")
    return n
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
# the fit's queue: the one the backward blend runs on
q_fit = collections.Counter(e[3] for e in ev if e[2].startswith("fused_blend_bwd")).most_common(1)[0][0]
main = [e for e in ev if e[3] == q_fit]
side = [e for e in ev if e[3] != q_fit]
starts = [i for i, e in enumerate(main) if e[2].startswith("fused_preprocess_bin") or e[2].startswith("fused_preprocess_fwd")]
its = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = main[a:b]
    names = [k[2] for k in ks]
    if not any(n.startswith("fused_blend_bwd") for n in names):
        continue                                   # a forward alone (evaluation render, trajectory frames)
    exact = names[0].startswith("fused_preprocess_fwd")
    staged = any(n.startswith("snapshot_stage") for n in names)
    snap_sync = any(n.startswith("center_blend") for n in names)
    kind = "looked-at (snapshot)" if (staged or snap_sync) else ("exact, plain" if exact else "reserved, plain")
    span = main[b][0] - ks[0][0]
    busy = sum(k[1] - k[0] for k in ks)
    other = sum(k[1] - k[0] for k in ks if not k[2].startswith(("fused_", "bin_", "ssim_", "loss_", "snapshot_stage", "center_", "rec_depth")))
    its.append((kind, span / 1e3, busy / 1e3, (span - busy) / 1e3, other / 1e3, len(ks), ks[0][0]))
by = collections.defaultdict(list)
for it in its:
    by[it[0]].append(it)
print(f"{len(its)} iterations on queue {q_fit}; side-queue kernels: {len(side)} ({sum(e[1] - e[0] for e in side) / 1e6:.1f} ms)")
for kind, v in by.items():
    a = np.array([x[1:6] for x in v])
    ok = a[:, 0] < 2000                            # (iterations with host work behind them -- densification, frame boundary -- apart)
    print(f"{kind:24s} n {len(v):5d}  span {np.median(a[ok, 0]):7.1f} us (mean {a[ok, 0].mean():7.1f})  kernels {np.median(a[ok, 1]):7.1f}  "
          f"idle {np.median(a[ok, 2]):6.1f} (mean {a[ok, 2].mean():6.1f})  foreign kernels {a[ok, 3].mean():5.1f}  launches {np.median(a[ok, 4]):.0f}"
          f"   | {int((~ok).sum())} long ones: {a[~ok, 0].sum() / 1e3:.1f} ms")
# what runs on the side queues, and how the fit's kernels fare while it does
busy_side = [(e[0], e[1]) for e in side if e[1] - e[0] > 5000]
def overlapped(k):
    return any(s < k[1] and e > k[0] for s, e in busy_side)
for name in ("fused_blend_fwd_kernel<0>", "fused_blend_bwd_kernel<10>", "fused_blend_bwd_kernel<7>", "loss_grad_kernel<false>",
             "fused_preprocess_bin_kernel<false>", "bin_tile_sort_kernel"):
    ks = [k for k in main if k[2] == name]
    if not ks:
        continue
    ov = np.array([overlapped(k) for k in ks])
    d = np.array([(k[1] - k[0]) / 1e3 for k in ks])
    if ov.any():
        print(f"{name:40s} alone {d[~ov].mean():6.1f} us (n {int((~ov).sum())})   beside a side-queue kernel {d[ov].mean():6.1f} us (n {int(ov.sum())})")
