"""Long clips: does the fused path drift away from the operator path over tens of frame boundaries?  (``-m gpu``)

The operator path (``fused=False``) is the reference's loop as it stands -- five msplat operators, autograd, torch Adam,
boolean gathers at the frame boundaries; the fused path is the native iteration with the frame-boundary state kept in
the engine.  Both are fitted to the SAME rigid synthetic clip (gflow_amd.synthetic._Scene) with the same seeds.  The two
optimisations are chaotic in the last bits (unordered LDS adds in the backward), so rows are not compared one by one:
what must agree is the quality of every frame, the number of splats and the still / moving labels."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

SMALL = dict(num_points=1500, iterations_first=60, iterations_after=40, iterations_camera=20, densify_interval=30,
             densify_times=1, densify_interval_after=20, densify_times_after=1, lambda_depth=1e-2)


def _fit_both(frames, cfg):
    from gflow_amd.fit_video import fit_clip
    out = []
    for fused in (True, False):
        keep = {}
        m = fit_clip(frames, DEV, cfg, seed=0, fused=fused, keep=keep)
        out.append((m, [float(p) for p in keep["psnr"]], keep["trainer"]))
    return out


@pytest.mark.parametrize("seed", [0, 1])
def test_twenty_four_frames_fused_and_operator_path_agree(seed):
    """24 frames at 96 x 128 with the occlusion-mask densification only (its weights are the mask itself, so both paths
    draw the SAME pixels from the same generator: the splat counts must be identical, and a label can differ only where
    the two optimisations left a splat on different sides of the move mask's edge)."""
    from gflow_amd import synthetic as S
    from gflow_amd.fit_video import upload_clip
    n = 24
    frames = upload_clip(S.make_clip(n, 96, 128, seed=seed, device=DEV), DEV)
    cfg = dict(SMALL, densify_interval=0, densify_interval_after=0)
    (ma, pa, ta), (mb, pb, tb) = _fit_both(frames, cfg)
    assert ma["iterations"] == mb["iterations"] == 60 + (n - 1) * 60
    assert ta.current_pts_num() == tb.current_pts_num() > 1500
    d = [x - y for x, y in zip(pa, pb)]
    # (ten fits sampled on one box, tools/drift_loop2.py: single frames up to 0.86 dB apart, mean -0.19 .. +0.07, second half
    #  -0.24 .. +0.10 -- two chaotic optimisations; the bounds of round 4, 0.9 / 0.25 / 0.35, sat on the largest of them)
    assert max(abs(v) for v in d) < 1.5, (pa, pb)                       # every frame
    assert abs(sum(d)) / n < 0.4, (pa, pb)                                # no systematic offset
    # ... and no drift: the second half of the clip is no further apart than the first
    assert abs(sum(d[n // 2:]) / (n - n // 2)) < 0.5, d
    differ = int((ta.still_mask != tb.still_mask).sum())
    assert differ <= 0.05 * ta.current_pts_num(), differ                  # (observed: 50 of 2192)
    assert abs(float(ta.still_mask.float().mean()) - float(tb.still_mask.float().mean())) < 0.02
    # both found the same camera (observed: <= 5.3e-3, the largest entry always the translation along the view axis,
    # the direction the photometric loss determines least)
    assert (ta.pose.detach() - tb.pose.detach()).abs().max().item() < 1.5e-2


def test_twenty_four_frames_with_error_densification():
    """The full recipe (error-map densification too: its draws depend on the error map, so the two paths append
    different splats): counts within 6 %, every frame's PSNR within 1.5 dB, no offset beyond 0.45 dB."""
    from gflow_amd import synthetic as S
    from gflow_amd.fit_video import upload_clip
    n = 24
    frames = upload_clip(S.make_clip(n, 96, 128, seed=0, device=DEV), DEV)
    (ma, pa, ta), (mb, pb, tb) = _fit_both(frames, SMALL)
    # (twelve runs on one box, tools/drift_loop.py: counts 0.1-3.1 % apart -- ~2 700 splats, the draws differ --, single frames
    #  up to 0.65 dB (1.09 seen in round 4), mean difference +0.01 .. +0.21 dB; the old bounds of 4 % and 0.3 dB were three
    #  standard deviations and failed one run in a dozen)
    assert abs(ta.current_pts_num() - tb.current_pts_num()) <= 0.06 * tb.current_pts_num(), (ta.current_pts_num(), tb.current_pts_num())
    d = [x - y for x, y in zip(pa, pb)]
    assert max(abs(v) for v in d) < 1.5 and abs(sum(d)) / n < 0.45, (max(abs(v) for v in d), sum(d) / n, pa, pb)


def test_eight_frames_at_480p_fused_and_operator_path_agree():
    """BASELINE configs[2]'s recipe at full size, eight frames, both paths: the drop from frame 0 to frame 1 (change of
    recipe: frozen colours, fewer iterations, lower rate) and the level of the later frames are the ALGORITHM's."""
    from gflow_amd import synthetic as S
    from gflow_amd.fit_video import upload_clip
    frames = upload_clip(S.make_clip(8, 480, 854, seed=0, device=DEV), DEV)
    (ma, pa, ta), (mb, pb, tb) = _fit_both(frames, dict(num_points=60000))
    assert ma["iterations"] == mb["iterations"] == 500 + 7 * 450
    assert max(abs(x - y) for x, y in zip(pa, pb)) < 0.8, (pa, pb)          # (observed: 0.1-0.5, run to run)
    assert abs(sum(x - y for x, y in zip(pa, pb))) / 8 < 0.4, (pa, pb)      # (observed: -0.17 .. +0.11 over six runs, tools/drift_loop.py 6 full)
    assert abs(ta.current_pts_num() - tb.current_pts_num()) <= 0.04 * tb.current_pts_num()   # (observed: up to 2.3 %, run to run)
    assert pa[0] - pa[1] > 0.8 and pb[0] - pb[1] > 0.8, (pa, pb)          # both paths show the step
