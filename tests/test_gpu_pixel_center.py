"""GFL_PIXEL_CENTER (include/gflow_hip.h), the tenth msplat-internal assumption of SURVEY.md 8c: where inside a pixel the
composite is sampled.  The shipped build uses 0 (pixel centres at integer coordinates).  Here the SAME sources are built a
second time with -DGFL_PIXEL_CENTER=0.5f (into a scratch directory; hipcc is on the GPU box) and a process of its own loads
that build (GFLOW_HIP_LIB) and holds it against the oracle with oracle.msplat_oracle.PIXEL_CENTER = 0.5: the five operators,
the fused render operator and its gradients, and the fused fit iteration (tests/test_gpu_fused.py's oracle-parity tests, called with the constant flipped) -- so that a maintainer who learns that upstream
samples at +0.5 flips ONE constant on each side and is covered by the same parity suite."""
import os
import shutil
import subprocess
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("xyz", "scale", "rotate", "opacity", "rgb")


def _probe():
    sys.path.insert(0, ROOT)
    import ctypes
    from gflow_amd import _lib
    import gflow_amd.render as R
    from oracle import msplat_oracle as MO
    from tests.scenes import random_scene
    from tests.test_gpu_parity import close_frac
    c = (ctypes.c_float * 16)()
    assert _lib.load().gfl_constants_n(c, 16) == 11 and c[10] == 0.5, list(c)
    MO.PIXEL_CENTER = 0.5
    dev = "cuda"
    s = random_scene(3000, 200, 136, seed=11, sigma_px=2.5)
    n, W, H = s["xyz"].shape[0], s["W"], s["H"]
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g))
    loss = lambda o, ww: (o["rgb"] * ww[0]).sum() + (o["depth_map"] * ww[1]).sum()
    leaves_c = {k: s[k].clone().requires_grad_(True) for k in NAMES}
    oc = MO.render_multiple([*[leaves_c[k] for k in NAMES], s["intr"], s["extr"], 0.33, W, H], ["rgb", "depth_map", "center"])
    loss(oc, w).backward()
    # a shift of half a pixel is far outside the tolerance: the default-centre oracle must NOT match (the test has teeth)
    MO.PIXEL_CENTER = 0.0
    with torch.no_grad():
        o0 = MO.render_multiple([*[s[k] for k in NAMES], s["intr"], s["extr"], 0.33, W, H], ["rgb"])
    MO.PIXEL_CENTER = 0.5
    assert ((o0["rgb"] - oc["rgb"].detach()).abs() > 1e-3).float().mean() > 0.2
    for fused in (False, True):
        R.USE_FUSED = fused
        leaves_g = {k: s[k].clone().to(dev).requires_grad_(True) for k in NAMES}
        group = [*[leaves_g[k] for k in NAMES], s["intr"].to(dev), s["extr"].to(dev), 0.33, W, H]
        og = R.render_multiple(group, ["rgb", "depth_map"])
        for k in ("rgb", "depth_map"):
            close_frac(og[k], oc[k].detach(), 1e-4, 1e-5, bad_frac=3e-4, hard=2e-2, what=f"{k} fused={fused}")
        loss(og, [t.to(dev) for t in w]).backward()
        for k in NAMES:
            ref = leaves_c[k].grad
            rel = (leaves_g[k].grad.cpu() - ref).norm() / ref.norm()
            assert rel < 2e-3, f"d_{k} (fused={fused}): relative L2 error {rel:.2e}"
    R.USE_FUSED = True
    with torch.no_grad():
        og = R.render_multiple([*[s[k].to(dev) for k in NAMES], s["intr"].to(dev), s["extr"].to(dev), 0.33, W, H], ["center"])
    close_frac(og["center"], oc["center"].detach(), 1e-4, 1e-5, bad_frac=3e-4, hard=2e-2, what="center")
    # the fused FIT iteration of the same build: the oracle-parity tests of tests/test_gpu_fused.py, called as functions with the
    # oracle's constant at 0.5 (forward, all gradients, three optimiser steps, the three snapshot images)
    from tests import test_gpu_fused as TF
    s2 = random_scene(2500, 168, 120, seed=31, sigma_px=2.5, tilt=False)
    setup = (s2, TF._raw_from_scene(s2), *TF._targets(s2["H"], s2["W"], 5))
    TF.test_fused_forward_matches_oracle_and_operator_path(setup)
    TF.test_fused_gradients_match_oracle(setup)
    TF.test_fused_three_steps_track_the_oracle(setup)
    TF.test_snapshot_images_match_the_oracle(setup)
    print("pixel-centre 0.5 build: operators, fused operator, gradients and the fused fit iteration match the oracle at 0.5")


pytestmark = pytest.mark.gpu


def test_a_build_with_pixel_centres_at_one_half_matches_the_oracle_at_one_half():
    with tempfile.TemporaryDirectory(prefix="gfl_pc_") as tmp:
        csrc = os.path.join(tmp, "gflow_amd", "csrc")
        os.makedirs(os.path.join(tmp, "include"))
        shutil.copytree(os.path.join(ROOT, "gflow_amd", "csrc"), csrc,
                        ignore=shutil.ignore_patterns("*.o", "*.so"))
        shutil.copy(os.path.join(ROOT, "include", "gflow_hip.h"), os.path.join(tmp, "include"))
        res = subprocess.run(["make", "-C", csrc, "-j16", "CONSTS=-DGFL_PIXEL_CENTER=0.5f"], capture_output=True, text=True,
                             timeout=900)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        lib = os.path.join(tmp, "gflow_amd", "libgflow_hip.so")
        assert os.path.exists(lib)
        env = dict(os.environ, GFLOW_HIP_LIB=lib)
        res = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, env=env, timeout=900)
        assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
        assert "match the oracle at 0.5" in res.stdout


if __name__ == "__main__":
    _probe()
