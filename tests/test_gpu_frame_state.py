"""Frame-boundary state against its restatement (``-m gpu``).

A three-frame clip is driven through ``SimpleGaussian.train`` the way fit_video does (first frame; then camera-only +
joint stage per later frame).  At BOTH boundaries the inputs the trainer saw are handed to oracle/frame_oracle.py --
the reference's boolean-gather formulation of trainer.py:347-376 (flow warp of the moving splats) and :588-625
(still / moving labels, ``last_*`` stash):
  * warped ``xyz`` to 1e-5, the rows that must NOT move bit-identical;
  * ``still_mask`` / ``still_mask_tentative`` / ``last_uv`` / ``last_num`` / ``last_xyz`` exactly.
The scene contains the cases that distinguish a right mask from a wrong one: splats that project outside the image and
behind the camera, splats whose pixel sits on the edge of the move mask, and rows appended by densification after the
previous boundary (label vectors shorter than the row count)."""
import pytest
import torch

from oracle import frame_oracle as FR

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, W, N0 = 96, 128, 1500


def _clip(n=3, seed=3):
    from gflow_amd import synthetic as S
    return S.make_clip(n, H, W, seed=seed)


def _edge_count(uv, move_mask):
    """splats inside the image whose pixel's label differs from one of its 4 neighbours (the move-mask edge)"""
    inside = (uv[:, 0] > 1) & (uv[:, 0] < W - 2) & (uv[:, 1] > 1) & (uv[:, 1] < H - 2)
    y, x = uv[inside][:, 1].long(), uv[inside][:, 0].long()
    c = move_mask[y, x]
    diff = (move_mask[y - 1, x] != c) | (move_mask[y + 1, x] != c) | (move_mask[y, x - 1] != c) | (move_mask[y, x + 1] != c)
    return int(diff.sum())


class _Spy:
    """records what make_stepper saw before its pre-update and what it left in ``xyz``"""

    def __init__(self, tr):
        self.tr, self.calls = tr, []
        self.orig = tr.make_stepper
        tr.make_stepper = self

    def __call__(self, **kw):
        tr = self.tr
        pre = dict(xyz=tr._attributes["xyz"].detach().clone().cpu(), camera_only=bool(kw.get("camera_only", False)),
                   has_still=hasattr(tr, "still_mask"))
        if pre["has_still"]:
            pre.update(last_uv=tr.last_uv.clone().cpu(), last_still_mask=tr.last_still_mask.clone().cpu(),
                       gt_flow=tr.gt_flow.clone().cpu(), gt_depth=tr.gt_depth.clone().cpu(),
                       intr=tr.intr.clone().cpu(), extr=tr.get_extr().detach().clone().cpu())
        st = self.orig(**kw)
        pre["xyz_after"] = tr._attributes["xyz"].detach().clone().cpu()
        self.calls.append(pre)
        return st


def _check_warp(call):
    want = FR.warp_moving(call["xyz"], call["last_uv"], call["last_still_mask"], call["gt_flow"], call["gt_depth"],
                          call["intr"], call["extr"], W, H)
    got = call["xyz_after"]
    M = call["last_still_mask"].shape[0]
    moving = ~call["last_still_mask"]
    inside = FR._inside(call["last_uv"][:M], W, H)
    moved = torch.zeros(got.shape[0], dtype=torch.bool)
    moved[:M] = moving & inside
    assert int(moved.sum()) > 10, "the scene has no moving splats inside the image"
    assert int((moving & ~inside).sum()) > 0, "no moving splat outside the image: the within test is not exercised"
    assert torch.equal(got[~moved], call["xyz"][~moved])                  # still rows, outside rows, appended rows
    assert torch.equal(want[~moved], call["xyz"][~moved])
    err = (got[moved] - want[moved]).abs().max().item()
    assert err < 1e-5, f"warped xyz differs from the restatement by {err:.2e}"
    assert (got[moved] - call["xyz"][moved]).abs().max().item() > 1e-4    # ... and they did move
    return int(moved.sum())


def _check_labels(tr, uv_engine, move_mask, last_still_before):
    n_now = tr.current_pts_num()
    still, tentative = FR.relabel(uv_engine.cpu(), move_mask, n_now, last_still_before)
    assert torch.equal(tr.still_mask.cpu(), still)
    assert torch.equal(tr.still_mask_tentative.cpu(), tentative)
    assert torch.equal(tr.last_still_mask.cpu(), still)
    assert torch.equal(tr.last_uv.cpu(), uv_engine.cpu())
    assert tr.last_num == n_now and tr.last_xyz.shape[0] == n_now
    assert torch.equal(tr.last_xyz.cpu(), tr.get_attribute("xyz").detach().cpu())
    return still, tentative


@pytest.mark.parametrize("fused", [True, False])
def test_frame_boundaries_match_the_restatement(fused):
    from gflow_amd.trainer import SimpleGaussian
    frames = _clip()
    f0 = frames[0]
    tr = SimpleGaussian(f0["image"], f0["depth"], num_points=N0, device=DEV, seed=0, fused=fused)
    tr.load_camera(focal=f0["focal"], pp=f0["pp"])
    tr.init_gaussians_from_image(f0["image"], f0["depth"], num_points=N0)
    # splats the labels must leave alone: 40 projecting right of / below the image (inside the 1.3 frustum margin, so
    # they do have a uv), 10 behind the camera (culled: depth 0, uv (0, 0))
    with torch.no_grad():
        xyz = tr._attributes["xyz"]
        z = xyz[:40, 2].clone()
        xyz[:20, 0] = (W + 6.0 - f0["pp"][0]) / f0["focal"] * z[:20]
        xyz[20:40, 1] = (H + 4.0 - f0["pp"][1]) / f0["focal"] * z[20:40]
        xyz[40:50, 2] = -1.0
    spy = _Spy(tr)
    common = dict(lambda_rgb=1.0, lambda_depth=1e-2, snapshot_interval=0)

    def engine_uv(n):
        if fused:
            return tr.engine.uv[:n].clone()
        return None

    # ---------------- frame 0
    tr.train(iterations=40, lr=4e-3, lambda_var=10.0, densify_interval=15, densify_times=1,
             move_mask=f0["move_mask"], **common)
    n_after0 = tr.current_pts_num()
    assert n_after0 > N0                                                       # densification appended rows
    uv0 = engine_uv(n_after0) if fused else tr.last_uv.clone()
    still0, _ = _check_labels(tr, uv0, f0["move_mask"], None)
    assert not bool(still0.all()) and bool(still0[:50].all())                   # the 50 outside / culled rows stay "still"
    assert _edge_count(uv0.cpu(), f0["move_mask"]) > 0
    assert not spy.calls[0]["has_still"] and torch.equal(spy.calls[0]["xyz"], spy.calls[0]["xyz_after"])

    # ---------------- frames 1, 2
    warped = []
    for i in (1, 2):
        fr = frames[i]
        # some MOVING splats are recorded outside the image at the boundary: the warp must skip them
        with torch.no_grad():
            mv = torch.nonzero(~tr.last_still_mask).flatten()[:25]
            tr.last_uv[mv, 0] = W + 3.0
        tr.set_gt_image(fr["image"]); tr.set_gt_depth(fr["depth"]); tr.set_gt_flow(frames[i - 1]["flow"])
        n_calls = len(spy.calls)
        tr.train(iterations=12, lr_camera=5e-4, lambda_var=0.0, lambda_still=0.0, lambda_flow=0.01, densify_interval=0,
                 camera_only=True, move_mask=fr["move_mask"], **common)
        cam_call = spy.calls[n_calls]
        assert cam_call["camera_only"] and torch.equal(cam_call["xyz"], cam_call["xyz_after"])   # no warp in this stage
        last_before = tr.last_still_mask.clone().cpu()
        n_before = tr.current_pts_num()
        tr.train(iterations=25, lr=1e-3, lr_camera=0.0, lambda_var=10.0, lambda_still=10.0, lambda_flow=0.01,
                 densify_interval=10, densify_times=1, densify_occ_percent=1.0, mask=fr["occ_mask"],
                 move_mask=fr["move_mask"], **common)
        warped.append(_check_warp(spy.calls[n_calls + 1]))
        n_now = tr.current_pts_num()
        assert n_now > n_before                                                 # occlusion-mask + error densification
        uv_i = engine_uv(n_now) if fused else tr.last_uv.clone()
        still, tentative = _check_labels(tr, uv_i, fr["move_mask"], last_before)
        assert last_before.shape[0] == n_before < n_now                         # the label vector was shorter than N
        assert torch.equal(still[:n_before], last_before)                       # old labels win (trainer.py:598-599)
        assert not torch.equal(still, tentative)                                # ... and differ from this frame's own
        assert _edge_count(uv_i.cpu(), fr["move_mask"]) > 0
    assert all(w > 10 for w in warped)


@pytest.mark.parametrize("fused", [True, False])
def test_mask_prompt_points_and_their_propagation_match_the_restatement(fused):
    """init_mask_prompt_pts (trainer.py:290-330; fit_video.py:155-157 calls it after the first frame with the first
    frame's segmentation) and the propagated mask of every later joint train() (trainer.py:611-619): the prompt's splat
    set exactly, the points handed to the hull exactly, and the hull mask itself -- built by gflow_amd.hull from exactly
    those points -- covering the moving disc it started from once the disc has moved."""
    import numpy as np
    from gflow_amd.hull import FastConcaveHull2D
    from gflow_amd.trainer import SimpleGaussian
    frames = _clip(3)
    f0 = frames[0]
    tr = SimpleGaussian(f0["image"], f0["depth"], num_points=N0, device=DEV, seed=0, fused=fused)
    tr.load_camera(focal=f0["focal"], pp=f0["pp"])
    tr.init_gaussians_from_image(f0["image"], f0["depth"], num_points=N0)
    common = dict(lambda_rgb=1.0, lambda_depth=1e-2, snapshot_interval=0)
    tr.train(iterations=40, lr=4e-3, lambda_var=10.0, densify_interval=15, densify_times=1, move_mask=f0["move_mask"], **common)
    assert not hasattr(tr, "propagate_seg")
    prompt = f0["move_mask"]                                         # the first frame's segmentation of the object
    pts = tr.init_mask_prompt_pts(prompt)
    # (the reference renders anew there: the projections of the rows as the LAST Adam step left them, not last_uv)
    with torch.no_grad():
        uv0 = tr.project_points(tr.get_attribute("xyz").detach())[0].cpu()
    assert torch.equal(pts.cpu(), FR.mask_prompt_points(uv0, prompt, W, H))
    assert 20 < int(pts.sum()) < pts.shape[0]
    for i in (1, 2):
        fr = frames[i]
        tr.set_gt_image(fr["image"]); tr.set_gt_depth(fr["depth"]); tr.set_gt_flow(frames[i - 1]["flow"])
        tr.train(iterations=15, lr_camera=5e-4, lambda_flow=0.01, camera_only=True, move_mask=fr["move_mask"], **common)
        tr.train(iterations=30, lr=1e-3, lambda_var=10.0, lambda_still=10.0, lambda_flow=0.01, densify_interval=20,
                 densify_times=1, mask=fr["occ_mask"], move_mask=fr["move_mask"], **common)
        uv = tr.last_uv.cpu()
        want_pts = FR.propagated_points(uv, pts.cpu(), W, H)
        assert want_pts.shape[0] > 4 and tr.mask_prompt_pts.shape[0] == pts.shape[0] < tr.current_pts_num()   # (rows were appended since)
        seg = tr.propagate_seg
        assert seg.dtype == np.uint8 and seg.shape == (H, W) and set(np.unique(seg)) <= {0, 255}
        want = (FastConcaveHull2D(want_pts.numpy()).mask(W, H) * 255).astype(np.uint8)
        assert np.array_equal(seg, want)
        # the propagated mask follows the object: it overlaps this frame's disc much more than the background
        gt = fr["move_mask"].numpy()
        assert (seg[gt] > 0).mean() > 0.6 and (seg[~gt] > 0).mean() < 0.15, ((seg[gt] > 0).mean(), (seg[~gt] > 0).mean())
