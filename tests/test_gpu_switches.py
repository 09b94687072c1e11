"""The library reads THREE environment switches, once per process (include/gflow_hip.h: GFL_EWA_MFMA, GFL_RESERVED,
GFL_FWD_SPLIT_MIN).  Each is flipped here in a process of its own -- this file run as a script: four fused iterations on a
seeded scene with a pile in one tile, what they left behind written to an .npz -- and held against the defaults:

* GFL_RESERVED=0 and GFL_FWD_SPLIT_MIN=<n> decide WHERE and in how many launches work is done, never what is computed:
  the FIRST forward's per-tile sorted lists and contributor counts bit for bit, its render and final T too (but for the
  long-tile walk, which forms its transmittance products in tree order: last bits on the tiles that take it); after four
  iterations everything within the spread of the backward's unordered LDS adds (which differ from run to run anyway: the
  rows of two runs of the SAME build differ in the last bits after one step, and later forwards with them).  That an
  iteration on reserved regions and one on the exact path produce bit-identical lists and renders from bit-identical rows is
  held by tests/test_gpu_fused.py::test_reserved_tile_regions_give_the_lists_of_the_exact_binning_path and, at 480p / 60 000
  splats, by tests/test_gpu_fullsize.py;
* GFL_EWA_MFMA=1 changes the order of three products per splat: same results to rounding.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(out_path):
    sys.path.insert(0, ROOT)
    from gflow_amd import _lib
    from tests.scenes import random_scene
    from tests.test_gpu_fused import POSE, _engine, _raw_from_scene, _targets
    torch.manual_seed(0)
    s = random_scene(4000, 168, 120, seed=77, sigma_px=2.5, tilt=False)
    # a pile: 700 small splats in one tile, so that the long-tile forward walk and the split sort have something to take
    n_pile = 700
    g = torch.Generator().manual_seed(3)
    f = s["intr"][0].item()
    z = 2.0 + 0.5 * torch.rand(n_pile, generator=g)
    u = 88.0 + 12.0 * torch.rand(n_pile, generator=g)
    v = 56.0 + 12.0 * torch.rand(n_pile, generator=g)
    s["xyz"][:n_pile] = torch.stack([(u - s["W"] / 2) / f * z, (v - s["H"] / 2) / f * z, z], dim=1)
    s["scale"][:n_pile] = (1.2 / f * z).unsqueeze(1) * (0.8 + 0.4 * torch.rand(n_pile, 3, generator=g))
    s["opacity"][:n_pile] = 0.05 + 0.1 * torch.rand(n_pile, 1, generator=g)
    raw = _raw_from_scene(s)
    img, dep = _targets(s["H"], s["W"], 5)
    eng = _engine(raw, s, img, dep, pose=POSE, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=2e-3, lr_camera=0.0,
                  total_iters=100)
    def left_behind(tag):
        torch.cuda.synchronize()
        tr = eng.tile_range.cpu().numpy()
        ids = eng.ids.cpu().numpy()
        lists = np.concatenate([ids[a:b] for a, b in tr] + [np.zeros(0, np.int32)])
        return {tag + "lists": lists, tag + "lens": tr[:, 1] - tr[:, 0], tag + "render": eng.render.cpu().numpy(),
                tag + "final_T": eng.final_T.cpu().numpy(), tag + "n_contrib": eng.n_contrib.cpu().numpy()}

    eng.forward()
    out = left_behind("f0_")
    for _ in range(4):
        eng.iteration()
    eng.check_overflow()
    out.update(left_behind(""))
    np.savez(out_path, params=eng.params[:eng.N].cpu().numpy(), sums=eng.sums.cpu().numpy(),
             mfma=np.int32(_lib.load().gfl_ewa_on_mfma()), longest=np.int32(out["lens"].max()), **out)


def _run(tmp_path, name, **env):
    out = str(tmp_path / f"{name}.npz")
    e = {k: v for k, v in os.environ.items() if k not in ("GFL_EWA_MFMA", "GFL_RESERVED", "GFL_FWD_SPLIT_MIN")}
    r = subprocess.run([sys.executable, os.path.abspath(__file__), out], env=dict(e, **env), cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return dict(np.load(out))


@pytest.fixture(scope="module")
def default_run(tmp_path_factory):
    return _run(tmp_path_factory.mktemp("switches"), "default")


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"GFL_RESERVED": "0"}, {"GFL_FWD_SPLIT_MIN": "100000"}, {"GFL_FWD_SPLIT_MIN": "128"}],
                         ids=lambda e: "-".join(f"{k}={v}" for k, v in e.items()))
def test_scheduling_switches_do_not_change_a_result(default_run, tmp_path, env):
    a, b = default_run, _run(tmp_path, "flipped", **env)
    assert a["longest"] > 448, "the scene must have a tile the default walks as four blocks"
    # the first forward: the same rows in both processes
    assert np.array_equal(a["f0_lens"], b["f0_lens"]) and np.array_equal(a["f0_lists"], b["f0_lists"])
    assert np.array_equal(a["f0_n_contrib"], b["f0_n_contrib"])
    if "GFL_FWD_SPLIT_MIN" in env:
        assert np.abs(a["f0_final_T"] - b["f0_final_T"]).max() <= 2e-6
        assert np.abs(a["f0_render"] - b["f0_render"]).max() <= 2e-5
        # (on this small image every tile is the first tile of a queue: any of them may take the long walk)
    else:
        assert np.array_equal(a["f0_final_T"], b["f0_final_T"]) and np.array_equal(a["f0_render"], b["f0_render"])
    # after four iterations: to the order of the backward's LDS adds
    _same_to_rounding(a, b)


def _same_to_rounding(a, b, rows=1e-5):
    assert np.abs(a["lens"].astype(np.int64) - b["lens"]).sum() <= 1e-3 * a["lens"].sum()
    err = np.abs(a["render"] - b["render"])
    assert (err > 1e-4 * np.maximum(np.abs(a["render"]), 1.0)).mean() < 1e-3 and err.max() < 5e-3, err.max()
    rel = np.linalg.norm(a["params"] - b["params"]) / np.linalg.norm(a["params"])
    assert rel < rows, rel
    assert np.allclose(a["sums"], b["sums"], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_matrix_core_contraction_gives_the_same_fit_to_rounding(default_run, tmp_path):
    a, b = default_run, _run(tmp_path, "mfma", GFL_EWA_MFMA="1")
    assert a["mfma"] == 0 and b["mfma"] == 1, "the switch did not reach the library"
    # last-bit differences of a conic can flip a (splat, tile) pair at the culling disc's edge: a handful of list entries
    _same_to_rounding(a, b, rows=1e-4)


if __name__ == "__main__":
    _probe(sys.argv[1])
