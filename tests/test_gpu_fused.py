"""Parity of the fused fit iteration (gfl_fit_forward / gfl_fit_backward_step through
gflow_amd.fused.FitEngine) against the CPU oracle's fit step and against the
operator-by-operator path (``-m gpu``).

Gradients of the fused path are read back from Adam's first moment after ONE step from
a zero state: m = (1 - beta1) * g exactly, so g = m / 0.1.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fit_oracle as FO
from oracle import loss_oracle as LO
from oracle import msplat_oracle as MO
from tests.scenes import random_scene
from tests.test_gpu_parity import close_frac

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _raw_from_scene(s):
    """Raw (pre-activation) parameters whose activations reproduce the scene."""
    return dict(xyz=s["xyz"], scale=s["scale"] * torch.where(torch.rand_like(s["scale"]) > 0.3, 1.0, -1.0),
                rotate=s["rotate"] * 1.7, opacity=torch.logit(s["opacity"].clamp(0.02, 0.98)) / 10.0,
                rgb=torch.logit(s["rgb"].clamp(0.02, 0.98)))


def _targets(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(H, W, 3, generator=g)
    dep = 1.0 + 3.0 * torch.rand(H, W, 1, generator=g)
    return img, dep


def _engine(raw, s, img, dep, pose=None, **hyper):
    from gflow_amd.fused import FitEngine
    eng = FitEngine(s["W"], s["H"], capacity=max(2 * raw["xyz"].shape[0], 1024), device=DEV, bg=hyper.pop("bg", 0.0))
    eng.set_splats({k: v.to(DEV) for k, v in raw.items()})
    eng.intr.copy_(s["intr"].to(DEV))
    if pose is not None:
        eng.pose.copy_(pose.to(DEV))
    eng.set_targets(img, dep if hyper.get("lambda_depth", 0) > 0 else None)
    for k, v in hyper.items():
        setattr(eng.hp, k, v)
    eng.reset_optimizer()
    return eng


POSE = torch.tensor([0.02, -0.03, 0.01, 0.99, 0.05, -0.02, 0.08])


@pytest.fixture(scope="module")
def setup():
    s = random_scene(2500, 168, 120, seed=31, sigma_px=2.5, tilt=False)
    raw = _raw_from_scene(s)
    img, dep = _targets(s["H"], s["W"], 5)
    return s, raw, img, dep


def test_fused_forward_matches_oracle_and_operator_path(setup):
    import gflow_amd.render as R
    s, raw, img, dep = setup
    eng = _engine(raw, s, img, dep, pose=POSE, bg=0.2)
    eng.forward()
    eng.check_overflow()
    act = FO.activate(raw)
    extr = LO.pose_to_extr(POSE)
    oc = MO.render_multiple([*act, s["intr"], extr, 0.2, s["W"], s["H"]], ["rgb", "depth_map", "uv", "depth"])
    ref4 = torch.cat([oc["rgb"], oc["depth_map"]])
    close_frac(eng.render, ref4, 1e-4, 1e-5, bad_frac=3e-4, hard=2e-2, what="fused render vs oracle")
    close_frac(eng.uv, oc["uv"], 1e-5, 1e-3, what="uv")
    close_frac(eng.depth, oc["depth"], 1e-6, 1e-6, what="depth")
    np.testing.assert_allclose(eng.extr.cpu().numpy().reshape(3, 4), extr.numpy(), atol=1e-6)
    # operator-by-operator HIP path on the same inputs
    og = R.render_multiple([*[a.to(DEV) for a in act], s["intr"].to(DEV), extr.to(DEV), 0.2, s["W"], s["H"]],
                           ["rgb", "depth_map"])
    api4 = torch.cat([og["rgb"], og["depth_map"]])
    close_frac(eng.render, api4, 2e-5, 2e-6, bad_frac=1e-4, hard=2e-2, what="fused vs operator path")
    # the exact-disc culling only ever drops pairs: K_fused <= K_api, same image
    vis = oc["depth"] != 0
    tiles = MO.ewa_project(act[0], MO.compute_cov3d(act[1], act[2], vis), s["intr"], extr, oc["uv"], s["W"], s["H"],
                           vis)[2]
    assert 0 < eng.K <= int(tiles.sum())
    # determinism
    r1 = eng.render.clone()
    ids1 = eng.ids[:eng.K].clone()
    eng.forward()
    assert torch.equal(r1, eng.render) and torch.equal(ids1, eng.ids[:eng.K])


def test_fused_gradients_match_oracle(setup):
    s, raw, img, dep = setup
    lam = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0)
    eng = _engine(raw, s, img, dep, pose=POSE, lr=1e-3, lr_camera=1e-3, total_iters=100, **lam)
    eng.iteration()
    eng.check_overflow()
    # oracle
    rc = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    pose = POSE.clone().requires_grad_(True)
    ab = torch.tensor([1.0, 0.0], requires_grad=True)
    frame = dict(image=img, depth=dep)
    loss, info = FO.fit_loss(rc, pose, ab, s["intr"], frame, 0.0, lam["lambda_rgb"], lam["lambda_depth"], lam["lambda_var"])
    loss.backward()
    l_rgb, l_depth = eng.loss_terms()
    assert abs(l_rgb.item() - info["l_rgb"].item()) <= 1e-4 * abs(info["l_rgb"].item())
    assert abs(l_depth.item() - info["l_depth"].item()) <= 1e-4 * abs(info["l_depth"].item())
    from gflow_amd.fused import COLS
    n = raw["xyz"].shape[0]
    g_all = eng.adam_m[:n] / (1.0 - 0.9)
    for k, (a, b) in COLS.items():
        ref = rc[k].grad.reshape(n, b - a)
        got = g_all[:, a:b].cpu()
        rel = (got - ref).norm() / ref.norm()
        assert rel < 2e-3, f"d_{k}: relative L2 error {rel:.2e}"
        close_frac(got, ref, 5e-3, 5e-4 * ref.abs().max().item(), bad_frac=1e-2, what=f"d_{k}")
    gp = (eng.pose_m / 0.1).cpu()
    rel = (gp - pose.grad).norm() / pose.grad.norm()
    assert rel < 2e-3, f"d_pose: relative L2 error {rel:.2e}  {gp} vs {pose.grad}"
    gab = (eng.ab_m / 0.1).cpu()
    np.testing.assert_allclose(gab.numpy(), ab.grad.numpy(), rtol=2e-3)
    assert int(eng.step.item()) == 1


def test_fused_adam_update_matches_torch_adam(setup):
    """Given the fused path's own gradients, the parameter update equals torch.optim.Adam
    + LinearLR (trainer.py:153,384) over several steps."""
    s, raw, img, dep = setup
    eng = _engine(raw, s, img, dep, pose=POSE, lr=4e-3, lr_camera=1e-3, total_iters=4, lambda_rgb=1.0,
                  lambda_depth=0.1, lambda_var=10.0)
    n = raw["xyz"].shape[0]
    p = eng.params[:n, :14].clone().cpu().requires_grad_(True)
    ref = torch.optim.Adam([p], lr=4e-3)
    sch = torch.optim.lr_scheduler.LinearLR(ref, start_factor=1.0, end_factor=0.1, total_iters=4)
    for it in range(6):
        m_before = eng.adam_m[:n, :14].clone()
        eng.iteration()
        g = ((eng.adam_m[:n, :14] - 0.9 * m_before) / 0.1).cpu()        # the gradient this step used
        p.grad = g
        ref.step(); sch.step()
    err = (eng.params[:n, :14].cpu() - p.detach()).abs().max().item()
    assert err < 5e-5, f"Adam trajectories diverge by {err:.2e}"


def test_fused_three_steps_track_the_oracle(setup):
    """Three full iterations (render, losses, backward, Adam + LinearLR, pose and depth affine)
    against the CPU oracle's OracleFit.  Adam's update is +-lr for any gradient well above eps,
    so parameters agree to a small fraction of lr except where a gradient is at rounding level."""
    s, raw, img, dep = setup
    lr = 1e-3
    eng = _engine(raw, s, img, dep, lr=lr, lr_camera=lr, total_iters=10, lambda_rgb=1.0, lambda_depth=0.1,
                  lambda_var=10.0)
    fit = FO.OracleFit(raw, s["intr"], dict(image=img, depth=dep), lr=lr, iterations=10, lambda_rgb=1.0,
                       lambda_depth=0.1, lambda_var=10.0, lr_camera=lr)
    for _ in range(3):
        eng.iteration()
        fit.step()
    from gflow_amd.fused import COLS
    n = raw["xyz"].shape[0]
    for k, (a, b) in COLS.items():
        d = (eng.params[:n, a:b].cpu() - fit.raw[k].detach().reshape(n, b - a)).abs()
        assert (d > 0.25 * lr).double().mean().item() < 0.03, f"{k}: {(d > 0.25 * lr).double().mean().item():.3f} off"
    assert (eng.pose.cpu() - fit.pose.detach()).abs().max().item() < 0.25 * lr
    assert (eng.depth_ab.cpu() - fit.depth_ab.detach()).abs().max().item() < 0.25 * lr


def test_fused_regularisers_and_masks(setup):
    """flow / still terms and the gradient-control flags (trainer.py:505-551)."""
    s, raw, img, dep = setup
    n = raw["xyz"].shape[0]
    g = torch.Generator().manual_seed(9)
    act = FO.activate(raw)
    extr = LO.pose_to_extr(POSE)
    uv0, _ = MO.project_point(act[0], s["intr"], extr, s["W"], s["H"])
    last_uv = uv0 + torch.randn(n, 2, generator=g)
    gt_flow = torch.randn(s["H"], s["W"], 2, generator=g)
    still = torch.rand(n, generator=g) > 0.5
    last_xyz = raw["xyz"] + 0.01 * torch.randn(n, 3, generator=g)
    W, H = s["W"], s["H"]
    and_mask = (last_uv[:, 0] > 0) & (last_uv[:, 0] < W - 1) & (last_uv[:, 1] > 0) & (last_uv[:, 1] < H - 1) & ~still
    yx = last_uv.long()
    gt_f = gt_flow[yx[:, 1].clamp(0, H - 1), yx[:, 0].clamp(0, W - 1)]
    lam_flow, lam_still = 0.01, 10.0
    eng = _engine(raw, s, img, dep, pose=POSE, lr=1e-3, lambda_rgb=1.0, lambda_flow=lam_flow, lambda_still=lam_still,
                  freeze_rgb=1)
    eng.set_regularisers(flow_target=last_uv + gt_f, flow_w=and_mask.float() / (2.0 * and_mask.sum()),
                         still_target=last_xyz, still_w=still.float() / still.sum(), row_flags=still.to(torch.uint8))
    eng.iteration()
    rc = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    pose = POSE.clone().requires_grad_(True)
    ab = torch.tensor([1.0, 0.0])
    loss, info = FO.fit_loss(rc, pose, ab, s["intr"], dict(image=img, depth=dep), 0.0, 1.0, 0.0, 0.0)
    loss = loss + lam_flow * LO.flow_loss(info["uv"], last_uv, gt_flow, and_mask) \
        + lam_still * LO.still_loss(rc["xyz"], last_xyz, still)
    loss.backward()
    from gflow_amd.fused import COLS
    g_all = (eng.adam_m[:n] / 0.1).cpu()
    assert torch.all(g_all[:, 11:14] == 0)                          # rgb frozen
    assert torch.all(g_all[still][:, 0:3] == 0)                     # xyz of still splats frozen
    ref_xyz = rc["xyz"].grad.clone()
    ref_xyz[still] = 0
    rel = (g_all[:, 0:3] - ref_xyz).norm() / ref_xyz.norm()
    assert rel < 2e-3, f"xyz gradient with flow+still terms: {rel:.2e}"
    for k in ("scale", "rotate", "opacity"):
        a, b = COLS[k]
        ref = rc[k].grad.reshape(n, b - a)
        rel = (g_all[:, a:b] - ref).norm() / ref.norm()
        assert rel < 2e-3, f"d_{k}: {rel:.2e}"
    # camera-only: every splat gradient is zero, the pose still gets one
    eng2 = _engine(raw, s, img, dep, pose=POSE, lr=1e-3, lr_camera=1e-3, lambda_rgb=1.0, freeze_all_splats=1)
    before = eng2.params[:n].clone()
    eng2.iteration()
    assert torch.equal(before, eng2.params[:n]) and eng2.pose_m.abs().sum() > 0


@pytest.mark.parametrize("stage", ["first_frame", "joint", "camera_only"])
def test_fused_scale_term_matches_oracle(setup, stage):
    """lambda_scale (trainer.py:495-502) inside the fused iteration: mean of |scale| / depth over the rows inside the
    image -- narrowed to the moving rows in the joint stage and to the still rows in the camera-only stage for the rows
    that carry a label (the reference's within_index aliases valid_uv_index)."""
    s, raw, img, dep = setup
    n = raw["xyz"].shape[0]
    lam_scale = 0.5
    g = torch.Generator().manual_seed(4)
    still = None
    hyper = dict(lr=1e-3, lr_camera=1e-3, lambda_rgb=1.0, lambda_scale=lam_scale)
    if stage != "first_frame":
        still = torch.rand(n - 300, generator=g) > 0.4            # the last 300 rows were appended later: no label
    if stage == "camera_only":
        hyper["freeze_all_splats"] = 1
    eng = _engine(raw, s, img, dep, pose=POSE, **hyper)
    if still is not None:
        flags = torch.zeros(n, dtype=torch.uint8)
        flags[:still.shape[0]] = still.to(torch.uint8) | 2
        eng.set_regularisers(row_flags=flags)
    eng.iteration()
    rc = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    pose = POSE.clone().requires_grad_(True)
    ab = torch.tensor([1.0, 0.0])
    loss, info = FO.fit_loss(rc, pose, ab, s["intr"], dict(image=img, depth=dep), 0.0, 1.0, 0.0, 0.0)
    l_scale = LO.scale_loss(torch.abs(rc["scale"]), info["uv"], info["depth"], s["W"], s["H"], still,
                            camera_only=(stage == "camera_only"))
    # the term alone first: its gradient is what the test is about, the render's is checked elsewhere
    g_scale_only = torch.autograd.grad(lam_scale * l_scale, [rc["scale"], rc["xyz"], pose], retain_graph=True)
    assert g_scale_only[0].abs().sum() > 0 and g_scale_only[1].abs().sum() > 0
    (loss + lam_scale * l_scale).backward()
    gp = (eng.pose_m / 0.1).cpu()
    rel = (gp - pose.grad).norm() / pose.grad.norm()
    assert rel < 2e-3, f"d_pose with the scale term: {rel:.2e}"
    if stage == "camera_only":
        return                                                    # splat gradients are zeroed in this stage
    from gflow_amd.fused import COLS
    g_all = (eng.adam_m[:n] / 0.1).cpu()
    ref_xyz = rc["xyz"].grad.clone()
    if still is not None:
        ref_xyz[:still.shape[0]][still] = 0
    for k, ref in (("scale", rc["scale"].grad), ("xyz", ref_xyz)):
        a, b = COLS[k]
        rel = (g_all[:, a:b] - ref).norm() / ref.norm()
        assert rel < 2e-3, f"d_{k} with the scale term: {rel:.2e}"
    # and the term is really in there: without it the scale gradient is measurably different
    eng0 = _engine(raw, s, img, dep, pose=POSE, lr=1e-3, lr_camera=1e-3, lambda_rgb=1.0)
    eng0.iteration()
    a, b = COLS["scale"]
    assert ((eng0.adam_m[:n, a:b] - eng.adam_m[:n, a:b]).norm() / eng.adam_m[:n, a:b].norm()).item() > 1e-3


def test_snapshot_images_match_the_oracle(setup):
    """gfl_fit_snapshot: rgb, depth_map_color and center of the last forward as uint8 images, made on the device after
    the backward (trainer.py:573-582 takes them every 10th iteration) -- against the oracle's render_multiple and
    render2img (render.py:76-106,158-166).  The two extra images are composites over the lists of the REAL footprints
    (exact-disc culled): a unit blob may reach a tile its nearly transparent splat does not, hence the small allowance."""
    s, raw, img, dep = setup
    eng = _engine(raw, s, img, dep, pose=POSE, lr=0.0, lr_camera=0.0, lambda_rgb=1.0, lambda_depth=0.1)
    eng.iteration()
    got = eng.snapshot().cpu().numpy()
    assert got.shape == (3, s["H"], s["W"], 3) and got.dtype == np.uint8
    act = FO.activate(raw)
    oc = MO.render_multiple([*act, s["intr"], LO.pose_to_extr(POSE), 0.0, s["W"], s["H"]], ["rgb", "depth_map_color", "center"])
    for k, name in enumerate(("rgb", "depth_map_color", "center")):
        want = (torch.clamp(oc[name].permute(1, 2, 0), 0.0, 1.0).numpy() * 255).astype(np.uint8)
        d = np.abs(got[k].astype(np.int32) - want.astype(np.int32))
        assert (d > 1).mean() < (1e-3 if name == "rgb" else 4e-3), f"{name}: {(d > 1).mean():.2e} of the values off"
    # the iteration after a snapshot is not disturbed by it (the snapshot reuses the forward's workspace)
    eng2 = _engine(raw, s, img, dep, pose=POSE, lr=1e-3, lambda_rgb=1.0, lambda_depth=0.1)
    eng3 = _engine(raw, s, img, dep, pose=POSE, lr=1e-3, lambda_rgb=1.0, lambda_depth=0.1)
    for _ in range(3):
        eng2.iteration(); eng2.snapshot()
        eng3.iteration()
    n = raw["xyz"].shape[0]
    assert (eng2.params[:n, :14] - eng3.params[:n, :14]).abs().max().item() < 1e-5


def test_snapshot_iteration_leaves_the_images_of_its_own_forward(setup):
    """gfl_fit_iteration_snapshot (FitEngine.iteration(snapshot=True)): the iteration whose forward composes rgb and
    depth_map_color in ONE walk and center with a kernel of its own, against the same iteration followed by gfl_fit_snapshot
    (which walks the lists again, after the backward): the three uint8 images byte for byte -- the second composite has the
    first one's alphas, transmittances and stop rule, term for term --, the render the loss sees bit for bit, and the rows
    and loss sums the iteration leaves (the extra sums change nothing the fit sees).  Twice, with and without a hipGraph."""
    s, raw, img, dep = setup
    n = raw["xyz"].shape[0]
    for use_graph in (False, True):
        a = _engine(raw, s, img, dep, pose=POSE, lr=1e-3, lr_camera=1e-3, lambda_rgb=1.0, lambda_depth=0.1)
        b = _engine(raw, s, img, dep, pose=POSE, lr=1e-3, lr_camera=1e-3, lambda_rgb=1.0, lambda_depth=0.1)
        for it in range(3):
            got = a.iteration(snapshot=True, use_graph=use_graph).clone()
            b.iteration(reserved=False)
            want = b.snapshot()
            for k, name in enumerate(("rgb", "depth_map_color", "center")):
                assert torch.equal(got[k], want[k]), f"{name}, iteration {it}, graph {use_graph}: " \
                    f"{(got[k] != want[k]).float().mean().item():.2e} of the bytes differ"
            assert torch.equal(a.render, b.render) and torch.equal(a.final_T, b.final_T) and torch.equal(a.n_contrib, b.n_contrib)
            # (the backward's LDS adds are unordered: two runs of the SAME iteration agree to the last bits only)
            rel = ((a.params[:n] - b.params[:n]).norm() / b.params[:n].norm()).item()
            assert rel < 1e-6 and torch.allclose(a.sums, b.sums, rtol=1e-5, atol=1e-8), (rel, a.sums, b.sums)
            _copy_engine_state(a, b)
    # ... and against the oracle's three images
    act = FO.activate(raw)
    eng = _engine(raw, s, img, dep, pose=POSE, lr=0.0, lr_camera=0.0, lambda_rgb=1.0, lambda_depth=0.1)
    got = eng.iteration(snapshot=True).cpu().numpy()
    oc = MO.render_multiple([*act, s["intr"], LO.pose_to_extr(POSE), 0.0, s["W"], s["H"]], ["rgb", "depth_map_color", "center"])
    for k, name in enumerate(("rgb", "depth_map_color", "center")):
        want = (torch.clamp(oc[name].permute(1, 2, 0), 0.0, 1.0).numpy() * 255).astype(np.uint8)
        d = np.abs(got[k].astype(np.int32) - want.astype(np.int32))
        assert (d > 1).mean() < (1e-3 if name == "rgb" else 4e-3), f"{name}: {(d > 1).mean():.2e} of the values off"


def test_trainer_fused_and_operator_paths_agree():
    """Short first-frame fit with densification through both trainer paths."""
    from gflow_amd import synthetic as S
    from gflow_amd.trainer import SimpleGaussian
    H, W, N = 96, 128, 1500
    frame = S.make_frame(H, W, seed=3)
    out = {}
    for fused in (True, False):
        tr = SimpleGaussian(frame["image"], frame["depth"], num_points=N, device=DEV, seed=0, fused=fused)
        tr.load_camera(focal=frame["focal"], pp=frame["pp"])
        tr.init_gaussians_from_image(frame["image"], frame["depth"], num_points=N)
        tr.train(iterations=24, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
                 move_mask=frame["move_mask"], densify_interval=12, densify_times=1, snapshot_interval=8,
                 log_interval=1)
        out[fused] = (tr.train_log, tr.current_pts_num(), tr.psnr().item())
    lf, lo = out[True][0], out[False][0]
    # identical first iteration; afterwards Adam's sign-like first steps amplify rounding-level
    # gradient differences (e.g. the rotation of still-isotropic splats), so the trajectories
    # only stay statistically close -- the reference itself is not reproducible run to run
    assert abs(lf[0]["rgb"] - lo[0]["rgb"]) <= 1e-4 * abs(lo[0]["rgb"]), (lf[0], lo[0])
    for i, tol in ((5, 0.05), (11, 0.08)):
        assert abs(lf[i]["rgb"] - lo[i]["rgb"]) <= tol * abs(lo[i]["rgb"]), (i, lf[i], lo[i])
    # densify_num = int(num_points * mask_ratio * percent): the error maps differ at rounding level
    assert abs(out[True][1] - out[False][1]) <= 3 and out[True][1] > N
    assert abs(out[True][2] - out[False][2]) < 1.0           # PSNR (dB) after 24 iterations
    assert lf[-1]["total"] < 0.5 * lf[0]["total"]


def test_fused_fullsize_matches_operator_path():
    from gflow_amd import synthetic as S
    import gflow_amd.render as R
    H, W, N = 480, 854, 60000
    frame = S.make_frame(H, W, seed=0)
    raw = S.init_splats(frame, N, seed=0, grown=True)
    s = dict(W=W, H=H, intr=raw["intr"])
    eng = _engine({k: raw[k] for k in ("xyz", "scale", "rotate", "opacity", "rgb")}, s, frame["image"], frame["depth"])
    eng.forward()
    eng.check_overflow()
    act = [a.to(DEV) for a in FO.activate(raw)]
    og = R.render_multiple([*act, raw["intr"].to(DEV), raw["extr"].to(DEV), 0.0, W, H], ["rgb", "depth_map"])
    api4 = torch.cat([og["rgb"], og["depth_map"]])
    close_frac(eng.render, api4, 2e-5, 2e-6, bad_frac=1e-4, hard=2e-2, what="480p/60k fused vs operator path")
    assert 60000 < eng.K < 4_000_000


@pytest.mark.parametrize("H,W,N", [(480, 854, 60000), (720, 1280, 150000)])
def test_fused_fullsize_gradients_match_operator_path(H, W, N):
    """480p / 60k splats: more tiles than queues, ~250 heavy tiles walked in segments in the
    backward pass, and -- second iteration -- the schedule built from the first iteration's
    measured work.  720p / 150k: 3600 tiles for 2048 resident workgroups, so the workgroups
    pull several tiles each through the per-queue counters.  With lr = 0 the parameters stay put,
    so after two iterations Adam's first moment is (0.1 + 0.09) * g."""
    from gflow_amd import losses
    from gflow_amd import synthetic as S
    import gflow_amd.render as R
    from gflow_amd.fused import COLS
    frame = S.make_frame(H, W, seed=0)
    raw = S.init_splats(frame, N, seed=0, grown=True)
    keys = ("xyz", "scale", "rotate", "opacity", "rgb")
    s = dict(W=W, H=H, intr=raw["intr"])
    lam = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0)
    eng = _engine({k: raw[k] for k in keys}, s, frame["image"], frame["depth"], lr=0.0, lr_camera=0.0, **lam)
    eng.iteration()
    m1 = eng.adam_m[:N].clone()
    eng.iteration()
    eng.check_overflow()
    m2 = eng.adam_m[:N].clone()
    # operator path: autograd through the five msplat operators and the loss kernels
    leaf = {k: raw[k].to(DEV).clone().requires_grad_(True) for k in keys}
    act = FO.activate(leaf)
    og = R.render_multiple([*act, raw["intr"].to(DEV), raw["extr"].to(DEV), 0.0, W, H], ["rgb", "depth_map"])
    render4 = torch.cat([og["rgb"], og["depth_map"]])
    ab = torch.tensor([1.0, 0.0], device=DEV)
    loss, _, _, _ = losses.image_loss(render4, frame["image"].to(DEV), frame["depth"].to(DEV), ab, 1.0, 0.1, None)
    loss = loss + lam["lambda_var"] * losses.var_loss(act[1])
    loss.backward()
    for k, (a, b) in COLS.items():
        ref = leaf[k].grad.reshape(N, b - a)
        for name, got in (("first", m1[:, a:b] / 0.1), ("second", m2[:, a:b] / 0.19)):
            rel = ((got - ref).norm() / ref.norm()).item()
            assert rel < 1e-3, f"d_{k} ({name} iteration): relative L2 error {rel:.2e}"
            close_frac(got.cpu(), ref.cpu(), 5e-3, 5e-4 * ref.abs().max().item(), bad_frac=1e-3, what=f"d_{k} {name}")


def test_footprint_mask_matches_the_extra_render(setup):
    """Camera-only stage: keep = ~(move_mask | (grey of a render of the moving splats alone > 0)),
    trainer.py:426-451.  The library marks the alpha >= 1/255 footprints instead of rendering."""
    import gflow_amd.render as R
    s, raw, img, dep = setup
    n = raw["xyz"].shape[0]
    g = torch.Generator().manual_seed(3)
    moving = torch.rand(n - 100, generator=g) < 0.02           # shorter than N, like still_mask_tentative
    move_mask = torch.zeros(s["H"], s["W"], dtype=torch.bool)
    move_mask[10:30, 40:90] = True
    eng = _engine(raw, s, img, dep, pose=POSE)
    eng.set_footprint_mask(move_mask, moving)
    eng.forward()
    keep = eng.keep.bool().cpu()
    act = [a.to(DEV) for a in FO.activate(raw)]
    sel = torch.zeros(n, dtype=torch.bool)
    sel[:moving.shape[0]] = moving
    extr = LO.pose_to_extr(POSE).to(DEV)
    og = R.render_multiple([*[a[sel.to(DEV)] for a in act], s["intr"].to(DEV), extr, 0.0, s["W"], s["H"]], ["rgb"])
    grey = 0.299 * og["rgb"][0] + 0.587 * og["rgb"][1] + 0.114 * og["rgb"][2]
    ref_keep = ~((grey > 0).cpu() | move_mask)
    mismatch = (keep != ref_keep).float().mean().item()
    assert mismatch <= 2e-4, f"footprint mask differs on {mismatch:.2e} of the pixels"
    assert 0.05 < (~ref_keep).float().mean().item() < 0.95          # the case is not degenerate
    # the mask is the running union over the iterations of the stage (trainer.py:451 rebinds move_mask inside
    # its loop): after a second forward with a shifted camera nothing that was masked comes back
    eng.pose[4] += 0.05
    eng.forward()
    keep2 = eng.keep.bool().cpu()
    assert not (keep2 & ~keep).any() and (keep & ~keep2).any()
    # the footprint's workgroups ride in the forward blend's launch (mode 3); a snapshot iteration keeps the footprint launch of
    # its own behind the blend: the same mask from the same rows and the same mask before it
    a = _engine(raw, s, img, dep, pose=POSE, lr=0.0, lr_camera=0.0)
    b = _engine(raw, s, img, dep, pose=POSE, lr=0.0, lr_camera=0.0)
    for e in (a, b):
        e.set_footprint_mask(move_mask, moving)
    a.iteration()
    b.iteration(snapshot=True)
    assert torch.equal(a.keep, b.keep) and torch.equal(a.keep.bool().cpu(), keep) and torch.equal(a.render, b.render)
    # a non-black background lights every pixel of the extra render: everything is masked
    eng.hp.bg = 0.5
    eng.set_footprint_mask(move_mask, moving)
    eng.forward()
    assert int(eng.keep.sum().item()) == 0


def test_fused_gradients_with_four_segment_heavy_tiles():
    """20 000 splats on 96x96 pixels: ~1100 entries per tile, so every tile is a "heaviest tile of
    its queue" (36 tiles < 256 queues) and the backward walks it in four checkpointed segments."""
    from gflow_amd import losses
    import gflow_amd.render as R
    from gflow_amd.fused import COLS
    s = random_scene(20000, 96, 96, seed=5, sigma_px=2.0, tilt=False)
    raw = _raw_from_scene(s)
    n = raw["xyz"].shape[0]
    img, dep = _targets(s["H"], s["W"], 9)
    eng = _engine(raw, s, img, dep, lr=0.0, lr_camera=0.0, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=0.0)
    eng.iteration()
    eng.check_overflow()
    lens = (eng.tile_range[:, 1] - eng.tile_range[:, 0])
    assert int(lens.max()) > 960, f"scene too sparse for four segments (longest list {int(lens.max())})"
    m1 = eng.adam_m[:n].clone()
    keys = ("xyz", "scale", "rotate", "opacity", "rgb")
    leaf = {k: raw[k].to(DEV).clone().requires_grad_(True) for k in keys}
    act = FO.activate(leaf)
    extr = LO.pose_to_extr(eng.pose.cpu()).to(DEV)
    og = R.render_multiple([*act, s["intr"].to(DEV), extr, 0.0, s["W"], s["H"]], ["rgb", "depth_map"])
    close_frac(eng.render, torch.cat([og["rgb"], og["depth_map"]]).detach(), 2e-5, 2e-6, bad_frac=1e-4, hard=2e-2,
               what="dense scene, fused vs operator path")
    ab = torch.tensor([1.0, 0.0], device=DEV)
    loss, _, _, _ = losses.image_loss(torch.cat([og["rgb"], og["depth_map"]]), img.to(DEV), dep.to(DEV), ab, 1.0, 0.1, None)
    loss.backward()
    for k, (a, b) in COLS.items():
        ref = leaf[k].grad.reshape(n, b - a)
        got = m1[:, a:b] / 0.1
        rel = ((got - ref).norm() / ref.norm()).item()
        assert rel < 2e-3, f"d_{k}: relative L2 error {rel:.2e}"


def test_fullsize_properties_sorted_lists_and_feature_linearity():
    """Size-independent properties at the benchmark size (480p, 60k splats), no oracle needed:
    every tile list is ordered by (depth, id); the list covers exactly the pairs counted; the blend
    is linear in the features (bg = 0) and reproduces itself."""
    from gflow_amd import synthetic as S
    import gflow_amd.msplat as msplat
    H, W, N = 480, 854, 60000
    frame = S.make_frame(H, W, seed=1)
    raw = S.init_splats(frame, N, seed=1, grown=True)
    s = dict(W=W, H=H, intr=raw["intr"])
    eng = _engine({k: raw[k] for k in ("xyz", "scale", "rotate", "opacity", "rgb")}, s, frame["image"], frame["depth"])
    eng.forward()
    eng.check_overflow()
    K = eng.K
    tr = eng.tile_range.long()
    lens = tr[:, 1] - tr[:, 0]
    assert int(lens.sum()) == K and int(lens.min()) >= 0
    ids = eng.ids[:K].long()
    depth = eng.depth.reshape(-1)[ids]
    tile_of = torch.repeat_interleave(torch.arange(tr.shape[0], device=DEV), lens)
    same_tile = tile_of[1:] == tile_of[:-1]
    d0, d1 = depth[:-1], depth[1:]
    ordered = (d1 > d0) | ((d1 == d0) & (ids[1:] > ids[:-1]))
    assert bool((ordered | ~same_tile).all()), "a tile list is not sorted by (depth, id)"
    assert bool((depth > 0).all()), "a culled splat was binned"
    # linearity of the operator in the features and run-to-run reproducibility of the forward
    rec = eng.rec[:N]
    uv, conic, op = rec[:, 0:2].contiguous(), rec[:, 2:5].contiguous(), rec[:, 5:6].contiguous()
    g = torch.Generator(device=DEV).manual_seed(0)
    f1, f2 = torch.rand(N, 3, device=DEV, generator=g), torch.rand(N, 3, device=DEV, generator=g)
    blend = lambda f: msplat.alpha_blending(uv, conic, op, f, eng.ids[:K], eng.tile_range, 0.0, W, H)
    a, b, c = blend(f1), blend(f2), blend(f1 + 2.0 * f2)
    assert (c - (a + 2.0 * b)).abs().max().item() < 2e-5
    assert torch.equal(blend(f1), a)
    first = eng.render.clone()
    eng.forward()
    assert torch.equal(eng.render, first), "the fused forward is not reproducible"


@pytest.mark.parametrize("H,W,N", [(48, 64, 800), (256, 272, 6000), (480, 854, 60000)])
def test_tile_schedule_is_a_partition_of_the_tiles(H, W, N):
    """Every tile sits in exactly one queue, before and after the scheduler has work feedback (12,
    272 and 1620 tiles against 256 queues: fewer tiles than queues, one partial extra round, many)."""
    from gflow_amd import synthetic as S
    frame = S.make_frame(H, W, seed=3)
    raw = S.init_splats(frame, N, seed=3, grown=True)
    s = dict(W=W, H=H, intr=raw["intr"])
    eng = _engine({k: raw[k] for k in ("xyz", "scale", "rotate", "opacity", "rgb")}, s, frame["image"], frame["depth"],
                  lr=1e-3, lambda_rgb=1.0, lambda_depth=0.1)
    T = eng.T
    for it in range(3):
        eng.iteration()
        for fwd in (True, False):                    # the forward and the backward blend have their own queues
            queues = eng.schedule(forward=fwd)
            tiles = torch.cat(queues)
            assert tiles.numel() == T, f"iteration {it}: {tiles.numel()} scheduled items for {T} tiles"
            assert torch.equal(torch.sort(tiles).values, torch.arange(T)), f"iteration {it}: not a partition"
    # with feedback the queues carry similar loads (weights = list lengths as a proxy here)
    if T >= 1024:
        lens = (eng.tile_range[:, 1] - eng.tile_range[:, 0]).cpu().float()
        loads = torch.tensor([float(lens[q].sum()) for q in queues])
        assert loads.max() < 1.6 * loads.mean()


@pytest.mark.parametrize("W,H,N", [(17, 9, 1), (33, 65, 7), (100, 40, 300), (16, 16, 50), (250, 31, 1200)])
def test_fused_iteration_on_odd_sizes_matches_operator_path(W, H, N):
    """Image sizes that are not multiples of the tile, one-tile images, a handful of splats: the
    fused forward equals the operator path and an iteration runs (queues shorter than the CU count,
    partial tiles, empty tiles)."""
    import gflow_amd.render as R
    s = random_scene(N, W, H, seed=W + H + N, sigma_px=2.0, tilt=False)
    raw = _raw_from_scene(s)
    img, dep = _targets(H, W, 3)
    eng = _engine(raw, s, img, dep, lr=1e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=1.0)
    eng.forward()
    act = [a.to(DEV) for a in FO.activate(raw)]
    extr = LO.pose_to_extr(eng.pose.cpu()).to(DEV)
    og = R.render_multiple([*act, s["intr"].to(DEV), extr, 0.0, W, H], ["rgb", "depth_map"])
    close_frac(eng.render, torch.cat([og["rgb"], og["depth_map"]]), 2e-5, 2e-6, bad_frac=1e-3, hard=2e-2,
               what=f"{W}x{H}, {N} splats")
    before = eng.params[:N].clone()
    for _ in range(3):
        eng.iteration()
    eng.check_overflow()
    assert torch.isfinite(eng.params[:N]).all() and torch.isfinite(eng.render).all()
    assert not torch.equal(eng.params[:N], before)
    assert sum(q.numel() for q in eng.schedule()) == eng.T


def test_iterations_that_do_not_move_the_camera_take_the_short_cut_with_the_same_result(setup):
    """lr_camera = 0 (first frame, joint stages): no camera launch, no pose gradient -- the loss sums, the depth affine and
    the step counter come from a workgroup of the backward blend launch (LossTail, gfl_fit_bwd.hip).  Against the same
    iterations with the gradient asked for (step_camera = 2: the camera launch as before): rows, moments, depth
    affine, loss sums and step counter agree (the two folds add the same partials in trees of different width), the
    pose does not move in either, and only the long way round reports d_extr."""
    s, raw, img, dep = setup
    lam = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0)
    fast = _engine(raw, s, img, dep, pose=POSE, lr=2e-3, lr_camera=0.0, total_iters=6, **lam)
    slow = _engine(raw, s, img, dep, pose=POSE, lr=2e-3, lr_camera=0.0, total_iters=6, **lam)
    slow.hp.step_camera = 2
    n = raw["xyz"].shape[0]
    for it in range(4):
        fast.iteration()
        slow.iteration()
        assert int(fast.step.item()) == int(slow.step.item()) == it + 1
        np.testing.assert_allclose(fast.sums[:5].cpu().numpy(), slow.sums[:5].cpu().numpy(), rtol=2e-6)
    assert torch.equal(fast.pose.cpu(), POSE) and torch.equal(slow.pose.cpu(), POSE)
    np.testing.assert_allclose(fast.depth_ab.cpu().numpy(), slow.depth_ab.cpu().numpy(), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(fast.ab_m.cpu().numpy(), slow.ab_m.cpu().numpy(), rtol=1e-4, atol=1e-9)
    assert not torch.equal(fast.depth_ab.cpu(), torch.tensor([1.0, 0.0]))          # ... and it did step
    err = (fast.params[:n, :14] - slow.params[:n, :14]).abs().max().item()
    assert err < 1e-6, f"rows differ by {err:.2e}"
    err_m = (fast.adam_m[:n, :14] - slow.adam_m[:n, :14]).abs().max().item()
    scale = slow.adam_m[:n, :14].abs().max().item()
    assert err_m < 1e-5 * scale
    assert float(slow.d_extr.abs().max()) > 0.0 and float(fast.d_extr.abs().max()) == 0.0
    # nothing is stepped any more after a densification (trainer.py:941-951): the depth affine stays, the counter runs
    fast.hp.step_camera = 0
    ab0 = fast.depth_ab.clone()
    fast.iteration()
    assert torch.equal(fast.depth_ab, ab0) and int(fast.step.item()) == 5


def test_snapshot_from_a_staged_copy_equals_the_snapshot_in_place(setup):
    """gfl_fit_snapshot_stage: what a forward left behind (records, ids, tile ranges, render, tile queues) copied into a second
    engine; gfl_fit_snapshot on the copy -- after the first engine has moved on by two iterations -- gives the images the
    first engine's own snapshot gave, bit for bit.  Mismatched engines are refused."""
    import ctypes
    from gflow_amd import _lib as L
    s, raw, img, dep = setup
    a = _engine(raw, s, img, dep, pose=POSE, lr=2e-3, lr_camera=0.0, lambda_rgb=1.0, lambda_depth=0.1)
    b = _engine(raw, s, img, dep, pose=POSE, lr=2e-3, lr_camera=0.0, lambda_rgb=1.0, lambda_depth=0.1)
    for _ in range(2):
        a.iteration()                                    # (the second iteration runs on queues built by the first)
    want = a.snapshot().clone()
    b.set_count(a.N)
    L.check(a.lib.gfl_fit_snapshot_stage(ctypes.byref(a.state()), ctypes.byref(b.state()), L.stream()), "stage")
    a.iteration(); a.iteration()                          # the source moves on: rows, records, lists and queues change
    got = b.snapshot()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert not torch.equal(a.snapshot(), want)
    b.set_count(a.N - 1)
    assert a.lib.gfl_fit_snapshot_stage(ctypes.byref(a.state()), ctypes.byref(b.state()), L.stream()) != 0


def test_degenerate_scenes_no_splat_one_splat_everything_culled(setup):
    """The ends of the range: an engine with NO rows, with ONE row, and with every splat behind the camera (nothing is
    binned: K = 0).  The render is the background, the iteration runs, the loss is that of the background image, and rows
    that render nothing keep their values except for what the variance regulariser does to their scales."""
    from gflow_amd.fused import FitEngine
    s, raw, img, dep = setup
    H, W = s["H"], s["W"]
    mse_bg = float(((0.2 - img) ** 2).mean())
    # ---- no rows at all
    eng = FitEngine(W, H, capacity=1024, device=DEV, bg=0.2)
    eng.intr.copy_(s["intr"].to(DEV))
    eng.set_targets(img, dep)
    eng.hp.lambda_rgb, eng.hp.lambda_depth, eng.hp.lr = 1.0, 0.1, 1e-3
    eng.reset_optimizer()
    assert eng.N == 0
    for _ in range(2):
        eng.iteration()
    eng.check_overflow()
    assert eng.K == 0 and int(eng.step.item()) == 2
    assert torch.allclose(eng.render, torch.full((4, H, W), 0.2, device=DEV))      # (the depth plane is blended over bg as well)
    assert bool(torch.isfinite(eng.d_render).all()) and bool(torch.isfinite(eng.sums).all())
    assert abs(float(eng.sums[0]) / (H * W) - mse_bg) < 1e-5 * mse_bg + 1e-7        # sums[0]: the per-pixel mse summed
    # ---- one row
    one = {k: v[:1].clone() for k, v in raw.items()}
    e1 = _engine(one, s, img, dep, pose=POSE, lr=1e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=1.0)
    p0 = e1.params[:1, :14].clone()
    e1.iteration(); e1.iteration()
    e1.check_overflow()
    assert bool(torch.isfinite(e1.params[:1]).all()) and not torch.equal(e1.params[:1, :14], p0)
    # ---- every splat behind the camera
    behind = {k: v.clone() for k, v in raw.items()}
    behind["xyz"][:, 2] = -behind["xyz"][:, 2].abs() - 1.0
    e2 = _engine(behind, s, img, dep, lr=1e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=0.0, bg=0.2)
    n = behind["xyz"].shape[0]
    q0 = e2.params[:n, :14].clone()
    e2.iteration(); e2.iteration()
    e2.check_overflow()
    assert e2.K == 0
    assert torch.allclose(e2.render[:3], torch.full((3, H, W), 0.2, device=DEV))
    assert torch.equal(e2.params[:n, :14], q0)                   # zero gradients, zero moments: Adam does not move a row
    assert float(e2.depth[:n].abs().max()) == 0.0                # culled: depth 0 (render.py:29)


def test_more_pairs_than_the_lists_hold_is_reported_not_silent(setup):
    """K_cap smaller than the number of (splat, tile) pairs: the scatter drops the pairs beyond it and raises the sticky
    overflow flag; ``check_overflow`` / ``watch_overflow`` + ``poll_overflow`` turn it into an error, nothing is written
    outside the lists (an iteration on the truncated lists still runs and stays finite), and an engine with room
    renders the same scene without complaint."""
    from gflow_amd.fused import FitEngine
    s, raw, img, dep = setup
    big = _engine(raw, s, img, dep, pose=POSE, lambda_rgb=1.0)
    big.forward()
    big.check_overflow()
    K = big.K
    assert K > 4000
    small = FitEngine(s["W"], s["H"], capacity=2 * raw["xyz"].shape[0], device=DEV, K_cap=K // 3)
    small.set_splats({k: v.to(DEV) for k, v in raw.items()})
    small.intr.copy_(s["intr"].to(DEV))
    small.pose.copy_(POSE.to(DEV))
    small.set_targets(img, None)
    small.hp.lambda_rgb, small.hp.lr = 1.0, 1e-3
    small.reset_optimizer()
    small.iteration()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(small.render).all()) and bool(torch.isfinite(small.params[:small.N]).all())
    with pytest.raises(RuntimeError, match="K_cap"):
        small.check_overflow()
    small.iteration()
    small.watch_overflow()                       # (non-blocking: the flag travels to pinned memory behind the queue)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="K_cap"):
        small.poll_overflow()


def _copy_engine_state(src, dst):
    n = src.N
    for a in ("params", "adam_m", "adam_v"):
        getattr(dst, a)[:n].copy_(getattr(src, a)[:n])
    for a in ("pose", "pose_m", "pose_v", "depth_ab", "ab_m", "ab_v", "step"):
        getattr(dst, a).copy_(getattr(src, a))


def test_four_iterations_in_one_call_track_four_single_iterations(setup):
    """gfl_fit_iterations(count = 4), launched one by one and replayed as ONE graph, against four single iterations (the
    iterations of a call but the first bin into reserved tile regions; the single ones do too, through FitEngine)."""
    s, raw, img, dep = setup
    hyper = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=4e-3, lr_camera=0.0, total_iters=100)
    a = _engine(raw, s, img, dep, pose=POSE, **hyper)
    b = _engine(raw, s, img, dep, pose=POSE, **hyper)
    for use_graph in (False, True, True):
        a.iteration(count=4, use_graph=use_graph)
        for _ in range(4):
            b.iteration()
    a.check_overflow()
    assert int(a.step.item()) == int(b.step.item()) == 12
    assert a.K == b.K
    rel = ((a.params[:a.N] - b.params[:b.N]).norm() / b.params[:b.N].norm()).item()
    assert rel < 1e-5, rel
    assert (a.render - b.render).abs().max().item() < 1e-3
    assert torch.allclose(a.rec[:a.N], b.rec[:b.N], rtol=1e-3, atol=1e-3)       # rec = the LAST forward's in both
    assert torch.allclose(a.depth_ab, b.depth_ab, rtol=1e-5, atol=1e-7) and torch.allclose(a.sums, b.sums, rtol=1e-4)
    # a moving camera: same entry, with the camera launch
    a.hp.lr_camera = 1e-3
    b.hp.lr_camera = 1e-3
    a.iteration(count=2)
    b.iteration(); b.iteration()
    assert torch.allclose(a.pose, b.pose, rtol=1e-4, atol=1e-6)


def test_overflowing_pair_lists_are_grown_and_the_skipped_iterations_run_again(setup):
    """K_cap smaller than the scene's pair count: every iteration's forward raises the sticky flag, and while it is set the
    library steps NOTHING (rows, moments, depth affine, step counter) and counts the iterations; settle_overflow doubles
    the lists and says how many to run again.  After that the engine is where an engine that never overflowed is."""
    from gflow_amd.fused import FitEngine
    s, raw, img, dep = setup
    hyper = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=2e-3, lr_camera=0.0, total_iters=50)
    big = _engine(raw, s, img, dep, pose=POSE, **hyper)
    big.forward()
    K = big.K
    small = FitEngine(s["W"], s["H"], capacity=big.cap, device=DEV, K_cap=K // 3)
    small.set_splats({k: v.to(DEV) for k, v in raw.items()})
    small.intr.copy_(s["intr"].to(DEV))
    small.pose.copy_(POSE.to(DEV))
    small.set_targets(img, dep)
    for k, v in hyper.items():
        setattr(small.hp, k, v)
    small.reset_optimizer()
    rows0 = small.params[:small.N].clone()
    for _ in range(3):
        small.iteration()
    torch.cuda.synchronize()
    assert torch.equal(small.params[:small.N], rows0) and int(small.step.item()) == 0       # nothing was stepped
    assert float(small.adam_m.abs().max()) == 0.0 and torch.equal(small.depth_ab.cpu(), torch.tensor([1.0, 0.0]))
    grown = 0
    while True:
        k = small.settle_overflow()
        if not k:
            break
        assert k == 3 or (grown >= 2 and 0 < k < 3)
        grown += 1
        for _ in range(k):
            small.iteration()
    # K/3 -> 2K/3 -> 4K/3 hold the lists; the tile regions the iterations after the first bin into (count + count / 4 + 32
    # positions per tile, include/gflow_hip.h) need one doubling more
    assert grown == small.pairs_grown and grown in (2, 3) and small.K_cap == (K // 3) << grown
    for _ in range(3):
        big.iteration()
    torch.cuda.synchronize()
    assert int(small.step.item()) == int(big.step.item()) == 3 and small.K == big.K
    rel = ((small.params[:small.N] - big.params[:big.N]).norm() / big.params[:big.N].norm()).item()
    assert rel < 1e-5, rel
    assert torch.allclose(small.depth_ab, big.depth_ab, rtol=1e-5, atol=1e-7)
    # the camera launch (a moving camera) skips and counts the same way
    small2 = FitEngine(s["W"], s["H"], capacity=big.cap, device=DEV, K_cap=K // 3)
    small2.set_splats({k: v.to(DEV) for k, v in raw.items()})
    small2.intr.copy_(s["intr"].to(DEV)); small2.pose.copy_(POSE.to(DEV)); small2.set_targets(img, dep)
    for k, v in hyper.items():
        setattr(small2.hp, k, v)
    small2.hp.lr_camera = 1e-3
    small2.reset_optimizer()
    small2.iteration(); small2.iteration()
    torch.cuda.synchronize()
    assert torch.equal(small2.pose.cpu(), POSE) and int(small2.step.item()) == 0 and small2.overflow[:2].tolist() == [1, 2]


# ---------------------------------------------------------------------------------------- reserved tile regions
def _lists(eng):
    """the sorted id list of every tile (the lists' positions in ``ids`` differ between the two binning paths)"""
    tr = eng.tile_range.cpu()
    ids = eng.ids.cpu()
    return [ids[int(a):int(b)] for a, b in tr.tolist()]


def _reserved_on(eng):
    import ctypes
    return eng.lib.gfl_fit_reserved_supported(ctypes.byref(eng.state()), ctypes.byref(eng.hp)) == 1


@pytest.mark.parametrize("with_scale_term", [False, True])
def test_reserved_tile_regions_give_the_lists_of_the_exact_binning_path(setup, with_scale_term):
    """An iteration that follows a full iteration bins into the tile regions that one's last launch reserved (one launch in
    place of preprocess + column scan + scatter).  Held against the exact path on the SAME rows: records, every tile's sorted
    list, the render and the transmittance bit for bit; the stepped rows to the order of the backward's LDS adds."""
    s, raw, img, dep = setup
    hyper = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=4e-3, lr_camera=0.0, total_iters=100)
    if with_scale_term:
        hyper["lambda_scale"] = 0.5
    a = _engine(raw, s, img, dep, pose=POSE, **hyper)
    b = _engine(raw, s, img, dep, pose=POSE, **hyper)
    if not _reserved_on(a):
        pytest.skip("reserved tile regions are switched off (GFL_RESERVED=0)")
    if with_scale_term:
        flags = (torch.arange(a.cap, device=DEV) % 3).to(torch.uint8)
        for e in (a, b):
            e.set_regularisers(row_flags=flags)
    a.iteration()                                    # exact (nothing reserved yet); its last launch reserves
    assert a._reserved_flag() == a.GFL_ITER_RESERVED
    _copy_engine_state(a, b)
    b.iteration(reserved=False)                      # iteration 1 on the exact path
    a.iteration()                                    # iteration 1 on the reserved regions
    a.check_overflow(); b.check_overflow()
    assert a.K == b.K > 0
    assert torch.equal(a.rec[:a.N], b.rec[:b.N])
    la, lb = _lists(a), _lists(b)
    assert all(torch.equal(x, y) for x, y in zip(la, lb))
    tr = a.tile_range.cpu()
    live = tr[:, 1] > tr[:, 0]
    assert bool((tr[live][1:, 0] >= tr[live][:-1, 1]).all())                 # regions in tile order, not overlapping
    assert int((tr[:, 1] - tr[:, 0]).sum()) == a.K and int(tr[:, 1].max()) > a.K      # ... with gaps between them
    assert torch.equal(a.render, b.render) and torch.equal(a.final_T, b.final_T) and torch.equal(a.n_contrib, b.n_contrib)
    for name in ("params", "adam_m", "adam_v"):
        x, y = getattr(a, name)[:a.N], getattr(b, name)[:b.N]
        assert (x - y).abs().max().item() <= 1e-5 * max(1.0, y.abs().max().item()), name
    assert int(a.step.item()) == int(b.step.item()) == 2
    # ten more, inside graphs and several per call: the two fits stay together
    for use_graph, count in ((False, 1), (True, 1), (True, 1), (True, 3), (True, 3), (False, 1)):
        a.iteration(use_graph=use_graph, count=count)
        for _ in range(count):
            b.iteration(reserved=False)
    a.check_overflow()
    assert int(a.step.item()) == int(b.step.item()) == 12 and a.K == b.K
    rel = ((a.params[:a.N] - b.params[:b.N]).norm() / b.params[:b.N].norm()).item()
    assert rel < 1e-5, rel
    assert (a.render - b.render).abs().max().item() < 1e-3


def test_a_tile_that_outgrows_its_reserved_region_voids_that_iteration_only(setup):
    """The regions are a prediction (count + count / 4 + 32 positions per tile).  Splats the host moves between two iterations
    -- here: a third of them onto one spot -- make tiles outgrow theirs: that iteration is void (nothing is stepped, it is
    counted), the regions reserved at its end are sized by what the tiles wanted, the next iteration is fine again, and
    settle_overflow says how many iterations to run in addition."""
    s, raw, img, dep = setup
    hyper = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=2e-3, lr_camera=0.0, total_iters=50)
    a = _engine(raw, s, img, dep, pose=POSE, **hyper)
    b = _engine(raw, s, img, dep, pose=POSE, **hyper)
    if not _reserved_on(a):
        pytest.skip("reserved tile regions are switched off")
    a.iteration(); b.iteration(reserved=False)
    mid = torch.tensor([s["W"] / 2.0, s["H"] / 2.0], device=DEV)
    centre = int((a.rec[:a.N, 0:2] - mid).norm(dim=1).argmin())
    for e in (a, b):
        e.params[0:e.N:3, 0:3] = e.params[centre, 0:3].clone()           # (in place: the engine does not know)
    rows = a.params[:a.N].clone()
    a.iteration()
    torch.cuda.synchronize()
    assert a.overflow[:3].tolist() == [0, 1, 1]                             # void, counted
    assert torch.equal(a.params[:a.N], rows) and int(a.step.item()) == 1
    a.check_overflow()                                                      # (not an error: nothing was lost)
    a.iteration()
    torch.cuda.synchronize()
    assert a.overflow[:3].tolist() == [0, 1, 0] and int(a.step.item()) == 2
    assert a.settle_overflow() == 1 and a.regions_outgrown == 1 and a.overflow.tolist() == [0, 0, 0, 0]
    a.iteration()
    b.iteration(reserved=False); b.iteration(reserved=False)
    assert a.settle_overflow() == 0
    assert int(a.step.item()) == int(b.step.item()) == 3 and a.K == b.K
    assert all(torch.equal(x, y) for x, y in zip(_lists(a), _lists(b)))
    rel = ((a.params[:a.N] - b.params[:b.N]).norm() / b.params[:b.N].norm()).item()
    assert rel < 1e-5, rel


def test_reserved_flag_without_reserved_regions_is_caught_on_the_device(setup):
    import ctypes
    from gflow_amd import _lib as L
    s, raw, img, dep = setup
    a = _engine(raw, s, img, dep, pose=POSE, lambda_rgb=1.0, lr=1e-3, lr_camera=0.0)
    if not _reserved_on(a):
        pytest.skip("reserved tile regions are switched off")
    L.check(a.lib.gfl_fit_iterations(ctypes.byref(a.state()), ctypes.byref(a.hp), 1, a.GFL_ITER_RESERVED, L.stream()), "it")
    with pytest.raises(RuntimeError, match="reserved"):
        a.check_overflow()
    assert int(a.step.item()) == 0


def test_reserved_tile_regions_on_a_720p_tile_grid():
    """3 600 tiles: the binning launch holds up to eight tiles' regions per lane, the region workgroup sixteen counts per
    lane (both sized for grids of up to 4 096 tiles); splats spread thin enough that most (block, tile) cells are empty."""
    s = random_scene(6000, 1280, 720, seed=5, sigma_px=6.0, tilt=False)
    raw = _raw_from_scene(s)
    img, dep = _targets(s["H"], s["W"], 7)
    hyper = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=2e-3, lr_camera=0.0, total_iters=100)
    a = _engine(raw, s, img, dep, pose=POSE, **hyper)
    b = _engine(raw, s, img, dep, pose=POSE, **hyper)
    if not _reserved_on(a):
        pytest.skip("reserved tile regions are switched off")
    assert a.T == 80 * 45
    a.iteration()
    _copy_engine_state(a, b)
    for _ in range(3):
        a.iteration()
        b.iteration(reserved=False)
    a.check_overflow(); b.check_overflow()
    assert a.overflow.tolist() == [0, 0, 0, 0]
    assert a.K == b.K > 0 and int(a.step.item()) == int(b.step.item()) == 4
    tr = a.tile_range.cpu()
    assert int(tr[:, 1].max()) > a.K                                         # (regions: gaps between the lists)
    rel = ((a.params[:a.N] - b.params[:b.N]).norm() / b.params[:b.N].norm()).item()
    assert rel < 1e-5, rel
    # the forward of the same rows: lists and render bit for bit
    _copy_engine_state(a, b)
    a.iteration(); b.iteration(reserved=False)
    assert all(torch.equal(x, y) for x, y in zip(_lists(a), _lists(b)))
    assert torch.equal(a.render, b.render) and torch.equal(a.rec[:a.N], b.rec[:b.N])


def test_a_void_iteration_inside_a_four_iteration_graph(setup):
    """The same inside one graph launch of four iterations: the first is void, the other three step, one is owed."""
    s, raw, img, dep = setup
    hyper = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, lr=2e-3, lr_camera=0.0, total_iters=50)
    a = _engine(raw, s, img, dep, pose=POSE, **hyper)
    b = _engine(raw, s, img, dep, pose=POSE, **hyper)
    if not _reserved_on(a):
        pytest.skip("reserved tile regions are switched off")
    a.iteration(); b.iteration(reserved=False)
    a.iteration(use_graph=True, count=4); a.iteration(use_graph=True, count=4)        # (captured, replayed once)
    for _ in range(8):
        b.iteration(reserved=False)
    mid = torch.tensor([s["W"] / 2.0, s["H"] / 2.0], device=DEV)
    centre = int((a.rec[:a.N, 0:2] - mid).norm(dim=1).argmin())
    for e in (a, b):
        e.params[0:e.N:3, 0:3] = e.params[centre, 0:3].clone()
    a.iteration(use_graph=True, count=4)
    torch.cuda.synchronize()
    assert a.overflow[:3].tolist() == [0, 1, 0] and int(a.step.item()) == 9 + 3
    assert a.settle_overflow() == 1
    a.iteration()
    for _ in range(4):
        b.iteration(reserved=False)
    assert int(a.step.item()) == int(b.step.item()) == 13 and a.K == b.K
    rel = ((a.params[:a.N] - b.params[:b.N]).norm() / b.params[:b.N].norm()).item()
    assert rel < 1e-5, rel
