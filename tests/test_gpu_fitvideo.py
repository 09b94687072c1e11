"""Multi-frame fits through gflow_amd.fit_video.fit_clip (``-m gpu``): exercises the frame-boundary
state (flow warp of moving splats, still/moving labels), the camera-only phase with the
tentative-moving footprint, the flow / still terms, occlusion-mask densification, and hipGraph
capture of the fused iteration."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

SMALL = dict(num_points=1500, iterations_first=60, iterations_after=40, iterations_camera=20, densify_interval=30,
             densify_times=1, densify_interval_after=20, densify_times_after=1, lambda_depth=1e-2)


def _clip(n=3, H=96, W=128, seed=0):
    from gflow_amd import synthetic as S
    return S.make_clip(n, H, W, seed=seed)


@pytest.mark.parametrize("fused", [True, False])
def test_three_frame_clip_runs_and_improves(fused):
    from gflow_amd.fit_video import fit_clip
    frames = _clip()
    logs = []
    m = fit_clip(frames, DEV, SMALL, seed=0, fused=fused, log=logs.append)
    assert m["frames"] == 3 and m["clips"] == 1
    assert m["iterations"] == 60 + 2 * (20 + 40)
    assert m["splats_final"] > 1500                    # densification appended splats
    assert m["psnr_sum"] / 3 > 20.0, logs              # every frame is fitted reasonably
    assert m["rasterisations"] >= m["iterations"]


def test_fused_and_operator_clips_reach_similar_quality():
    from gflow_amd.fit_video import fit_clip
    frames = _clip(seed=1)
    a = fit_clip(frames, DEV, SMALL, seed=0, fused=True)
    b = fit_clip(frames, DEV, SMALL, seed=0, fused=False)
    assert abs(a["psnr_sum"] - b["psnr_sum"]) / 3 < 1.5
    assert abs(a["splats_final"] - b["splats_final"]) <= 0.02 * b["splats_final"]


def test_camera_only_phase_moves_the_pose_not_the_splats():
    from gflow_amd import synthetic as S
    from gflow_amd.trainer import SimpleGaussian
    frames = _clip(2)
    f0, f1 = frames
    tr = SimpleGaussian(f0["image"], f0["depth"], num_points=1500, device=DEV, seed=0)
    tr.load_camera(focal=f0["focal"], pp=f0["pp"])
    tr.init_gaussians_from_image(f0["image"], f0["depth"], num_points=1500)
    tr.train(iterations=40, lr=4e-3, lambda_rgb=1.0, lambda_depth=1e-2, lambda_var=10.0, move_mask=f0["move_mask"],
             densify_interval=0, snapshot_interval=0)
    assert hasattr(tr, "still_mask") and tr.still_mask.dtype == torch.bool
    tr.set_gt_image(f1["image"]); tr.set_gt_depth(f1["depth"]); tr.set_gt_flow(f0["flow"])
    before = {k: v.clone() for k, v in tr._attributes.items()}
    pose0 = tr.pose.detach().clone()
    tr.train(iterations=15, lr_camera=1e-3, lambda_rgb=1.0, lambda_depth=1e-2, lambda_flow=0.01, camera_only=True,
             move_mask=f1["move_mask"], densify_interval=0, snapshot_interval=0)
    for k in before:
        assert torch.equal(before[k], tr._attributes[k]), k          # trainer.py:548-551
    assert (tr.pose.detach() - pose0).abs().max() > 0                # the camera did move
    assert (tr.pose.detach() - pose0).abs().max() < 15 * 1e-3 * 1.01  # at most lr_camera per Adam step


def test_fused_iteration_is_hipgraph_capturable():
    """The whole iteration allocates nothing and never reads back: capture it in a hipGraph,
    replay it, and get bit-identical parameters to eager launches."""
    from gflow_amd import synthetic as S
    from gflow_amd.fused import FitEngine
    H, W, N = 96, 128, 1500
    frame = S.make_frame(H, W, seed=2)
    raw = S.init_splats(frame, N, seed=2, grown=True)

    def make():
        eng = FitEngine(W, H, capacity=4096, device=DEV)
        eng.set_splats({k: raw[k] for k in ("xyz", "scale", "rotate", "opacity", "rgb")})
        eng.intr.copy_(raw["intr"].to(DEV))
        eng.set_targets(frame["image"], frame["depth"])
        eng.hp.lr, eng.hp.total_iters, eng.hp.lambda_depth, eng.hp.lambda_var = 4e-3, 50, 0.1, 10.0
        eng.reset_optimizer()
        return eng

    eager = make()
    eager.iteration()                                     # also loads every kernel before the capture
    graphed = make()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            graphed.iteration()                           # recorded, not executed
    torch.cuda.current_stream().wait_stream(s)
    assert int(graphed.step.item()) == 0
    g.replay()
    torch.cuda.synchronize()
    assert int(graphed.step.item()) == 1
    # the forward is bitwise deterministic; the backward sums the four waves of a tile with LDS
    # float atomics, whose order is free, so parameters agree to rounding, not bit for bit
    assert torch.equal(eager.render, graphed.render)
    a, b = eager.params[:N, :14], graphed.params[:N, :14]
    fin = torch.isfinite(a) & torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), torch.isfinite(b))
    assert (a[fin] - b[fin]).abs().max().item() < 1e-5
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert int(graphed.step.item()) == 4
    assert torch.isfinite(graphed.render).all()


def test_clip_read_back_from_disk_fits_like_the_in_memory_clip(tmp_path):
    """A clip laid out on disk in the reference's folder convention (gflow_amd/io.py) and read back
    fits to the same quality as the in-memory clip (the PNGs quantise the image to 8 bits)."""
    from gflow_amd import io as gio
    from gflow_amd.fit_video import fit_clip
    frames = _clip()
    seq = gio.write_sequence(frames, str(tmp_path / "clip"))
    disk = gio.load_sequence(seq, frame_range=len(frames))
    a = fit_clip(frames, DEV, SMALL, seed=0)
    b = fit_clip(disk, DEV, SMALL, seed=0)
    assert b["frames"] == 3 and b["iterations"] == a["iterations"]
    assert abs(a["psnr_sum"] - b["psnr_sum"]) / 3 < 1.0, (a, b)
