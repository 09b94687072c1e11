"""CPU oracle for the GFlow per-frame Gaussian-splatting hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gflow_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker.

PARITY UNPINNED: the rasteriser arithmetic of the reference lives in the
third-party CUDA extension ``msplat`` (github.com/pointrix-project/msplat), which
is not vendored under /root/reference, has no pinned version there, and cannot be
built or run here (CUDA only).  The reference has no tests or golden vectors for
this path.  The rasteriser oracle therefore restates the
published 3DGS / EWA-splatting algorithm at the reference's own call sites
(gflow/utils/render.py:21-105) and is pinned only by analytic known-answer tests;
the parts of the path that ARE importable from the reference (SSIM, pix2world,
turbo colour map) are pinned by golden vectors in tests/golden/.
"""
