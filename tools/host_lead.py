"""Is the host ahead of the device at the stage boundaries of a clip fit?  At the return of every stepper.run() and at the
first iteration of every stage: the host clock and an event on the fit's stream; lead = when the device gets there minus when
the host got there.  (analysis tool)      gpurun -- python tools/host_lead.py [frames] [snapshot_interval] [traj]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV, trainer as T

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
snap = int(sys.argv[2]) if len(sys.argv) > 2 else 10
traj = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0, device=dev), dev)
cfg = dict(num_points=60000, traj_num=traj, traj_offset=2)
FV.fit_clip(frames[:2], dev, cfg, seed=0, snapshot_interval=snap)
torch.cuda.synchronize()
marks = []
base = {}


def mark(label):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((label, time.perf_counter(), ev))


orig_make = T.SimpleGaussian.make_stepper


def make(self, *a, **k):
    mark("make_stepper begin cam=%s" % k.get("camera_only", False))
    st = orig_make(self, *a, **k)
    k_iters = k.get("iterations")
    mark("make_stepper end")
    run0 = st.fn_batch
    state = {"first": True}

    def run(n):
        if state["first"]:
            state["first"] = False
            mark("first run() call")
        run0(n)
        mark("run(%d) returned at it %d" % (n, st.iteration))
    st.fn_batch = run
    fin0 = st.settle

    def fin(*a, **k):
        fin0(*a, **k)
        if st.iteration >= (k_iters or 0):
            mark("settle returned")
    st.settle = fin
    return st


T.SimpleGaussian.make_stepper = make
from gflow_amd import fused as FU
orig_it = FU.FitEngine.iteration
orig_retire = FU.FitEngine._retire_graphs
orig_cb = torch.cuda.CUDAGraph.capture_begin
orig_ce = torch.cuda.CUDAGraph.capture_end
orig_rp = torch.cuda.CUDAGraph.replay
slow = []


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        dt = (time.perf_counter() - t0) * 1e3
        if dt > 0.3:
            slow.append((name, (t0 - t_b) * 1e3, dt))
        return r
    return w


FU.FitEngine.iteration = timed("FitEngine.iteration", orig_it)
FU.FitEngine._retire_graphs = timed("  _retire_graphs", orig_retire)
torch.cuda.CUDAGraph.capture_begin = timed("  capture_begin", orig_cb)
torch.cuda.CUDAGraph.capture_end = timed("  capture_end", orig_ce)
torch.cuda.CUDAGraph.replay = timed("  replay", orig_rp)
FU.FitEngine.state = timed("  state()", FU.FitEngine.state)
FU.FitEngine.read_pending = timed("read_pending", FU.FitEngine.read_pending)
FU.FitEngine.watch_pending = timed("watch_pending", FU.FitEngine.watch_pending)
orig_dens = T.SimpleGaussian.densify_by_pixels


def dens(self, *a, **k):
    mark("densify begin")
    r = orig_dens(self, *a, **k)
    mark("densify end")
    return r


T.SimpleGaussian.densify_by_pixels = dens
fs = torch.cuda.Stream(device=dev)
with torch.cuda.stream(fs):
    torch.cuda.synchronize()
    b_ev = torch.cuda.Event(enable_timing=True)
    b_ev.record()
    t_b = time.perf_counter()
    globals()['t_b'] = t_b
    m = FV.fit_clip(frames, dev, cfg, seed=0, snapshot_interval=snap)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_b
print(f"{n_frames} frames in {wall:.3f} s")
print("host calls that took more than 0.3 ms (start ms, duration ms), last 60:")
for name, t0, dt in slow[-60:]:
    print(f"   {name:24s} at {t0:9.2f} ms took {dt:7.2f} ms")
prev_h = prev_g = 0.0
for label, th, ev in marks:
    h = (th - t_b) * 1e3
    g = b_ev.elapsed_time(ev)
    print(f"{label:38s} host {h:9.2f} ms (+{h - prev_h:7.2f})   device {g:9.2f} ms (+{g - prev_g:7.2f})   device behind host by {g - h:8.2f} ms")
    prev_h, prev_g = h, g
