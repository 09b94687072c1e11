"""Pixel-error densification on the device (gflow_amd.trainer.densify_by_pixels) against a CPU
restatement of gflow/trainer.py:878-951 (oracle/densify_oracle.py), and the optimiser quirk it triggers (A13) against a
torch.optim sequence (``-m gpu``)."""
import numpy as np
import pytest
import torch

from oracle import densify_oracle as DO

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ("xyz", "scale", "rotate", "opacity", "rgb")


def _trainer(H=96, W=128, N=1500, seed=0, fused=True):
    from gflow_amd import synthetic as S
    from gflow_amd.trainer import SimpleGaussian
    frame = S.make_frame(H, W, seed=seed)
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=N, device=DEV, seed=seed, fused=fused)
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    tr.init_gaussians_from_image(frame["image"], frame["depth"], num_points=N)
    return tr, frame


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("with_mask", [False, True])
def test_new_splats_equal_the_restatement(fused, with_mask):
    """Count, and -- given the pixels that were drawn -- position, scale, colour, rotation and opacity of the appended
    rows are trainer.py:896-934; the old rows are untouched and stay where they were."""
    tr, frame = _trainer(fused=fused)
    tr.train(iterations=6, lr=4e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0, move_mask=frame["move_mask"],
             densify_interval=0, snapshot_interval=0)
    g = torch.Generator().manual_seed(1)
    err = (torch.rand(tr.H, tr.W, generator=g) ** 4).to(DEV)
    err[:10] = 0.0                                              # zero-error rows: only the uniform floor there
    mask = frame["occ_mask"] if with_mask else None
    thr, percent = (0.0, 0.7) if with_mask else (0.05, 0.4)
    before = {k: tr._attributes[k].detach().clone() for k in NAMES}
    extr = tr.get_extr().detach().cpu()
    n0, n1 = tr.densify_by_pixels(err if not with_mask else torch.ones_like(err), error_threshold=thr, percent=percent,
                                  mask=mask)
    p, ratio = DO.sampling_distribution((err if not with_mask else torch.ones_like(err)).cpu().numpy(), thr,
                                        None if mask is None else mask.numpy())
    assert n1 - n0 == DO.densify_num(tr.num_points, ratio, percent) > 0
    assert tr.current_pts_num() == n1
    idx = tr.last_densify_idx.cpu()
    assert idx.shape[0] == n1 - n0
    assert np.all(p.flatten()[idx.numpy()] > 0)                 # only pixels with probability were drawn
    ref = DO.new_splats(idx // tr.W, idx % tr.W, frame["image"], tr.gt_depth.cpu(), tr.intr.cpu(), extr, tr.num_points)
    for k in NAMES:
        got = tr._attributes[k].detach().cpu()
        assert torch.equal(got[:n0], before[k].cpu()), k        # old rows: bit for bit
        tail, want = got[n0:], ref[k]
        fin = torch.isfinite(want)
        assert torch.equal(torch.isfinite(tail), fin), k        # (logit(1) = inf for saturated pixels, as in the reference)
        np.testing.assert_allclose(tail[fin].numpy(), want[fin].numpy(), rtol=2e-5, atol=1e-6, err_msg=k)


def test_sampled_pixels_follow_the_masked_error_map():
    """chi-square of 400 000 draws against p = masked error / sum (np.random.choice(..., p=...), trainer.py:905)."""
    tr, frame = _trainer(H=48, W=64, N=500)
    g = torch.Generator().manual_seed(3)
    err = (torch.rand(48, 64, generator=g) ** 3)
    mask = torch.rand(48, 64, generator=g) > 0.35
    w, m = tr.densify_weights(err.to(DEV), 0.0, mask)
    p, ratio = DO.sampling_distribution(err.numpy(), 0.0, mask.numpy())
    np.testing.assert_allclose((w / w.sum()).cpu().numpy(), p, rtol=1e-5, atol=1e-9)
    assert abs(float(m.float().mean()) - ratio) < 1e-6
    n = 400000
    idx = tr.sample_pixels(w, n).cpu().numpy()
    obs = np.bincount(idx, minlength=p.size).astype(np.float64)
    exp = n * p.flatten()
    assert obs[exp == 0].sum() == 0
    sel = exp > 5
    chi2 = (((obs - exp) ** 2)[sel] / exp[sel]).sum()
    df = sel.sum() - 1
    assert abs(chi2 - df) < 6.0 * np.sqrt(2.0 * df), (chi2, df)


def test_densification_optimiser_quirk_matches_a_torch_sequence():
    """trainer.py:941-951: densification REPLACES the optimiser by Adam(attributes only, lr = self.lr).  So from then
    on the moments restart from zero, the lr stays at its initial value (the LinearLR keeps stepping the OLD
    optimiser), and the pose / depth-affine groups are never stepped again.  The fused path does this with flags;
    replay it with torch.optim on the gradients the fused iterations used."""
    tr, frame = _trainer()
    lr, lr_cam, iters, dens_at = 4e-3, 1e-3, 12, 5
    st = tr.make_stepper(iterations=iters, lr=lr, lr_camera=lr_cam, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
                         move_mask=frame["move_mask"], densify_interval=dens_at + 1, densify_times=1,
                         densify_err_thre=1e-3, densify_err_percent=0.3, snapshot_interval=0)
    tr.use_graph = False
    eng = tr.engine
    n0 = eng.N
    p = torch.nn.Parameter(eng.params[:n0, :14].clone())
    pose = torch.nn.Parameter(eng.pose.clone())
    ab = torch.nn.Parameter(eng.depth_ab.clone())
    opt = torch.optim.Adam([{"params": [p], "lr": lr}, {"params": [pose], "lr": lr_cam}, {"params": [ab], "lr": lr}])
    sch = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.1, total_iters=iters)
    grabbed = {}
    orig_reset = eng.reset_optimizer

    def spy_reset(splats=True, camera=True):
        grabbed["m"], grabbed["pose_m"], grabbed["ab_m"] = eng.adam_m.clone(), eng.pose_m.clone(), eng.ab_m.clone()
        return orig_reset(splats=splats, camera=camera)

    eng.reset_optimizer = spy_reset
    for it in range(iters):
        n_now = eng.N
        m_b, pm_b, am_b = eng.adam_m[:n_now, :14].clone(), eng.pose_m.clone(), eng.ab_m.clone()
        st()
        if it == dens_at:
            assert eng.N > n0 and "m" in grabbed                 # the event happened, moments were reset
            m_a, pm_a, am_a = grabbed["m"][:n_now, :14], grabbed["pose_m"], grabbed["ab_m"]
        else:
            m_a, pm_a, am_a = eng.adam_m[:n_now, :14], eng.pose_m, eng.ab_m
        p.grad = (m_a - 0.9 * m_b) / 0.1
        if it <= dens_at:
            pose.grad, ab.grad = (pm_a - 0.9 * pm_b) / 0.1, (am_a - 0.9 * am_b) / 0.1
        opt.step()
        if it <= dens_at:
            sch.step()
        if it == dens_at:
            # the reference's new optimiser: attributes only (old rows + appended rows), constant lr, fresh state
            p = torch.nn.Parameter(torch.cat([p.detach(), eng.params[n0:eng.N, :14].clone()]))
            opt = torch.optim.Adam([p], lr=lr)
            pose_frozen, ab_frozen = pose.detach().clone(), ab.detach().clone()
    err = (eng.params[:eng.N, :14] - p.detach()).abs()
    fin = torch.isfinite(p.detach())
    assert err[fin].max().item() < 5e-5, f"parameter trajectories diverge by {err[fin].max().item():.2e}"
    assert torch.allclose(eng.pose, pose_frozen, atol=1e-6) and torch.allclose(eng.depth_ab, ab_frozen, atol=1e-6)
    assert (pose_frozen - torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=DEV)).abs().max() > 1e-4
    assert int(eng.step.item()) == iters - dens_at - 1           # the step counter restarted with the new optimiser
