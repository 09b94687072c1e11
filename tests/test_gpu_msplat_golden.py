"""The HIP path against a capture of REAL msplat (tests/golden/msplat_ref.npz, tools/capture_msplat_golden.py): skipped while
nobody has made one; and the same comparison on a capture of THIS repository's operators -- a format check of script and
test, not a pin (``-m gpu``)."""
import os
import subprocess
import sys

import pytest

from tests import msplat_golden as G

pytestmark = pytest.mark.gpu


def _fused(leaves, intr, extr, bg, W, H):
    import gflow_amd.render as R
    return R.render(leaves, dict(intr=intr, extr=extr, W=W, H=H), bg)


def test_hip_path_matches_real_msplat():
    import gflow_amd.msplat as msplat
    if not os.path.exists(G.REF_PATH):
        pytest.skip("no capture of real msplat: run tools/capture_msplat_golden.py on a box where `import msplat` works")
    g, meta = G.load()
    assert meta["is_reference"], f"tests/golden/msplat_ref.npz was captured from {meta['module']!r}, not from msplat"
    G.check_complete(g, meta)
    for line in G.hold(msplat, g, meta, "cuda", fused_render=_fused):
        print(line)


def test_capture_script_runs_against_this_repositorys_operators(tmp_path):
    """FORMAT CHECK: the capture script with ``--module gflow_amd.msplat`` (what it will do with ``msplat`` on a CUDA box),
    then the oracle AND the HIP operators / fused operator held to that file."""
    import gflow_amd.msplat as msplat
    from oracle import msplat_oracle as MO
    out = tmp_path / "format_check.npz"
    res = subprocess.run([sys.executable, os.path.join(G.ROOT, "tools", "capture_msplat_golden.py"), "--module",
                          "gflow_amd.msplat", "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    g, meta = G.load(str(out))
    assert meta["is_reference"] is False
    G.check_complete(g, meta)
    for line in G.hold(MO, g, meta, "cpu"):                                  # the oracle against the HIP capture
        print("oracle vs HIP capture --", line)
    for line in G.hold(msplat, g, meta, "cuda", fused_render=_fused):        # HIP (deterministic forward) against itself + fused
        print("HIP vs HIP capture --", line)
