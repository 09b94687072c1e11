"""Is there a gap between consecutive graph replays?  One iteration per graph against two / four per graph.
    gpurun -- python tools/graph_gap.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gflow_amd import synthetic as S, _lib as L
from gflow_amd.trainer import SimpleGaussian

dev = torch.device("cuda", 0)
frame = S.make_frame(bench.H, bench.W, seed=0)
raw = S.init_splats(frame, bench.N_SPLATS, seed=0, grown=True)
tr = SimpleGaussian(frame["image"], frame["depth"], num_points=bench.N_SPLATS, device=dev, seed=0)
tr.load_camera(focal=frame["focal"], pp=frame["pp"])
for k in ("xyz", "scale", "rotate", "opacity", "rgb"):
    tr._attributes[k] = raw[k].to(dev)
stepper = tr.make_stepper(iterations=100000, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
                          move_mask=frame["move_mask"], densify_interval=0, snapshot_interval=0)
for _ in range(100):
    stepper()
eng = tr.engine
torch.cuda.synchronize()
for per in (1, 2, 4, 8, 1, 2, 4, 8):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(per):
                L.check(eng.lib.gfl_fit_iteration(ctypes.byref(eng.state()), ctypes.byref(eng.hp), L.stream()), "capture")
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 400 // per
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{per} iteration(s) per graph: {dt / (n * per) * 1e3:.4f} ms per iteration")
