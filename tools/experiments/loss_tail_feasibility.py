"""VERDICT r04 item 3 / DESIGN section 8 item 7: the SSIM statistics launch inside the forward blend's idle tail, gated by
per-tile completion counters that stay inside the producer's XCD (an agent-scope release on gfx950 writes back the XCD's whole
L2).  What this probe measures on the bench scene before anything is built:
  (1) how many statistics items (tile, channel) COULD run there: a tile's 16 x 16 outputs need the render of the tile and of
      its eight neighbours (5-pixel halo) -- all nine must be tiles of the same XCD band of the forward's schedule;
  (2) how much the forward's launch leaves idle (per-CU end times need a TRACE build: tools/bwd_trace.py --fwd);
  (3) an upper bound for ANY overlap of the loss pair with the forward phase: gfl_fit_forward on one stream, the loss pair (on
      the previous render) on another, against the two one after the other.
    gpurun -- python tools/experiments/loss_tail_feasibility.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gflow_amd import synthetic as S, _lib as L
from gflow_amd.fused import FitEngine

H, W, N = 480, 854, 60000
dev = "cuda"
frame = S.make_frame(H, W, seed=0)
raw = S.init_splats(frame, N, seed=0, grown=True)
eng = FitEngine(W, H, capacity=2 * N, device=dev)
eng.set_splats({k: raw[k].to(dev) for k in ("xyz", "scale", "rotate", "opacity", "rgb")})
eng.intr.copy_(raw["intr"].to(dev))
eng.set_targets(frame["image"], frame["depth"])
for k, v in dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=4e-3, lr_camera=0.0, total_iters=500).items():
    setattr(eng.hp, k, v)
eng.reset_optimizer()
for _ in range(32):
    eng.iteration()
torch.cuda.synchronize()

# ---- (1) bands of the forward's schedule
queues = eng.schedule(forward=True)
gx, gy = eng.gx, eng.gy
band = torch.full((gy, gx), -1, dtype=torch.long)
for q, tiles in enumerate(queues):
    for t in tiles.tolist():
        band[t // gx, t % gx] = q % 8
assert int((band < 0).sum()) == 0
pad = torch.nn.functional.pad(band[None, None].float(), (1, 1, 1, 1), mode="replicate")[0, 0].long()
same = torch.ones(gy, gx, dtype=torch.bool)
for dy in (0, 1, 2):
    for dx in (0, 1, 2):
        same &= pad[dy:dy + gy, dx:dx + gx] == band
print(f"tiles {gx * gy}; tiles whose 3 x 3 neighbourhood lies in ONE XCD band: {int(same.sum())} = {float(same.float().mean()):.3f}")
rows_per_band = [int((band == b).any(dim=1).sum()) for b in range(8)]
print("tile rows touched by band 0..7:", rows_per_band, f"(the image has {gy} tile rows)")

# ---- (3) forward || loss pair on two streams
lib = L.load()
ws = torch.empty(int(lib.gfl_loss_workspace_bytes(W, H)), dtype=torch.uint8, device=dev)
d_render = torch.empty_like(eng.render); err = torch.empty(H, W, device=dev); sums = torch.empty(8, device=dev)
render_old = eng.render.clone()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def loss(stream):
    L.check(lib.gfl_loss_fwd_bwd(L.ptr(render_old), L.ptr(eng.gt_rgb), L.ptr(eng.gt_depth), None, L.ptr(eng.depth_ab), 1.0, 0.1, W, H,
                                 L.ptr(d_render), L.ptr(err), L.ptr(sums), L.ptr(ws), ws.numel(), ctypes.c_void_p(stream.cuda_stream)), "loss")


def fwd(stream):
    L.check(lib.gfl_fit_forward(ctypes.byref(eng.state()), ctypes.byref(eng.hp), ctypes.c_void_p(stream.cuda_stream)), "fwd")


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def serial():
    fwd(sa); loss(sa)


def overlapped():
    fwd(sa); loss(sb)


t_f, t_l = timed(lambda: fwd(sa)), timed(lambda: loss(sa))
t_s, t_o = timed(serial), timed(overlapped)
print(f"forward phase {t_f:.1f} us, loss pair (uncached target statistics, with its fold) {t_l:.1f} us, one after the other {t_s:.1f} us, "
      f"on two streams {t_o:.1f} us: at most {t_s - t_o:.1f} us of the loss pair hide beside the WHOLE forward phase (binning, sort, blend)")
