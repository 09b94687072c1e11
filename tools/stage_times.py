"""Device time of every stage of a clip fit, per iteration, and of the pauses between the stages (events on the fit's stream at
the first run() call and at the end of every train() call's iterations).   (analysis tool)
    gpurun -- python tools/stage_times.py [frames] [snapshot_interval] [traj]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gflow_amd import synthetic as S, fit_video as FV, trainer as T

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
snap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
traj = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tail = None      # (round 6's "exact tail" experiment is gone: tools/experiments/README.md)
dev = torch.device("cuda", 0)
frames = FV.upload_clip(S.make_clip(n_frames, 480, 854, seed=0, device=dev), dev)
cfg = dict(num_points=60000, traj_num=traj, traj_offset=2)
init = T.SimpleGaussian.__init__


def patched(self, *a, **k):
    init(self, *a, **k)
    if tail is not None:
        self.exact_tail = tail


T.SimpleGaussian.__init__ = patched
FV.fit_clip(frames[:2], dev, cfg, seed=0, snapshot_interval=snap)
torch.cuda.synchronize()
stages = []
orig_make = T.SimpleGaussian.make_stepper


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def make(self, *a, **k):
    st = orig_make(self, *a, **k)
    rec = {"cam": bool(k.get("camera_only", False)), "iters": k.get("iterations"), "N0": self.current_pts_num(), "e0": None}
    run0, fin0 = st.fn_batch, st.settle

    def run(n):
        if rec["e0"] is None:
            rec["e0"] = ev()
        run0(n)

    def fin(*a, **k):
        # (the look at the end of the call: train_steps calls settle() once its iterations are queued)
        last_look = st.iteration >= rec["iters"] and "e1" not in rec
        if last_look:
            rec["e1"] = ev()
        fin0(*a, **k)
        if last_look:
            rec["N1"] = self.current_pts_num()
            stages.append(rec)
    st.fn_batch, st.settle = run, fin
    return st


T.SimpleGaussian.make_stepper = make
fs = torch.cuda.Stream(device=dev)
with torch.cuda.stream(fs):
    torch.cuda.synchronize()
    e_begin = ev()
    t0 = time.perf_counter()
    m = FV.fit_clip(frames, dev, cfg, seed=0, snapshot_interval=snap)
    e_end = ev()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
tot_it = sum(s["e0"].elapsed_time(s["e1"]) for s in stages)
print(f"{n_frames} frames, snapshots {snap}, traj {traj}, exact_tail {tail}: wall {wall*1e3:.1f} ms, {m['iterations']} iterations = "
      f"{wall / m['iterations'] * 1e6:.1f} us each; inside the stages' iteration loops {tot_it:.1f} ms, outside {e_begin.elapsed_time(e_end) - tot_it:.1f} ms")
prev = e_begin
for j, s in enumerate(stages):
    d = s["e0"].elapsed_time(s["e1"])
    gap = prev.elapsed_time(s["e0"])
    prev = s["e1"]
    if j < 5 or j % 10 in (1, 2) or j >= len(stages) - 4:
        print(f"stage {j:3d} {'camera' if s['cam'] else 'splats'} {s['iters']:4d} it  N {s['N0']:6d}->{s['N1']:6d}  {d:7.2f} ms = {d / s['iters'] * 1e3:6.1f} us/it   pause before it {gap:6.2f} ms")
cam = [s for s in stages[1:] if s["cam"]]
jnt = [s for s in stages[1:] if not s["cam"]]
for name, ss in (("camera", cam), ("joint", jnt)):
    if ss:
        print(f"{name}: mean {sum(s['e0'].elapsed_time(s['e1']) for s in ss) / sum(s['iters'] for s in ss) * 1e3:.1f} us/it over {len(ss)} stages")
pauses = []
prev = e_begin
for s in stages:
    pauses.append(prev.elapsed_time(s["e0"]))
    prev = s["e1"]
pauses.append(prev.elapsed_time(e_end))
print(f"pauses between stages: total {sum(pauses):.1f} ms, mean {sum(pauses[1:-1]) / max(len(pauses) - 2, 1):.2f} ms, first {pauses[0]:.1f}, last {pauses[-1]:.1f}")
