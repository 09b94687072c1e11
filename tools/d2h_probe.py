"""Does a device-to-host copy on a second stream hold up the kernels of the main stream?  (The snapshot images of a
clip fit leave for page-locked memory every tenth iteration, 3.7 MB; rocprofv3 shows the kernel that runs meanwhile taking
as long as the copy: tools/snapshot_timeline.sh.)
    gpurun -- python tools/d2h_probe.py [small]       # try with HSA_ENABLE_SDMA=..., GPU_FORCE_BLIT_COPY_SIZE=..., ...
A chain of kernels that keeps the device busy (64 MB each, ~25 us; `small`: 4 MB, latency-bound like the column scan),
alone and with a 3.7 MB copy enqueued on another stream every 12 kernels."""
import os, sys, time
import torch

dev = torch.device("cuda", 0)
small = "small" in sys.argv
x = torch.zeros((1 << 20) if small else (1 << 23), device=dev)
src = torch.zeros(3, 480, 854, 3, dtype=torch.uint8, device=dev)
pin = torch.empty(16, 3, 480, 854, 3, dtype=torch.uint8).pin_memory()
side = torch.cuda.Stream(device=dev)
main = torch.cuda.Stream(device=dev)
N = 192


def body(copies):
    for i in range(N):
        x.add_(1.0)
        if copies and i % 12 == 6:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pin[(i // 12) % 16].copy_(src, non_blocking=True)


def chain(copies, graph=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(main):
        if graph is not None:
            graph.replay()
            if copies:
                with torch.cuda.stream(side):
                    for k in range(16):
                        pin[k].copy_(src, non_blocking=True)
        else:
            body(copies)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(3):
    chain(False); chain(True)
a = min(chain(False) for _ in range(5))
b = min(chain(True) for _ in range(5))
# the chain as ONE graph (no launch gaps at all), the sixteen copies enqueued beside it
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    x.add_(1.0)
    with torch.cuda.graph(g, stream=main):
        for i in range(N):
            x.add_(1.0)
for _ in range(2):
    chain(False, g); chain(True, g)
ga = min(chain(False, g) for _ in range(5))
gb = min(chain(True, g) for _ in range(5))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(16):
    pin[i].copy_(src, non_blocking=True)
torch.cuda.synchronize()
c = (time.perf_counter() - t0) * 1e3 / 16
print(f"env {[(k, v) for k, v in os.environ.items() if k.startswith(('HSA_E', 'GPU_', 'DEBUG_CLR', 'ROC_'))]}")
print(f"{N} kernels alone {a:.3f} ms; with 16 copies on a second stream {b:.3f} ms: +{(b - a) / 16 * 1e3:.1f} us per copy; "
      f"as a graph {ga:.3f} / {gb:.3f} ms: +{(gb - ga) / 16 * 1e3:.1f} us per copy; a copy alone {c * 1e3:.1f} us")
