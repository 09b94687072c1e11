"""gflow_amd -- MI355X-native per-frame Gaussian-splatting optimiser for GFlow.

Hot path only: the msplat-compatible rasteriser operators
(`gflow_amd.msplat`), the render orchestrator (`gflow_amd.render`), the loss /
Adam kernels (`gflow_amd.losses`, `gflow_amd.optim`) and the trainer / fit_video
loop mirrors (`gflow_amd.trainer`, `gflow_amd.fit_video`).  All device work goes
through the C ABI of libgflow_hip.so (include/gflow_hip.h); there is no CPU path.
"""
__version__ = "0.1.0"
