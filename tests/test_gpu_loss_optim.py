"""HIP-vs-oracle parity of the fused loss, Adam and colour-map kernels (``-m gpu``).
The loss oracle IS pinned to the reference (tests/test_oracle_golden.py); the SSIM
golden vectors are also replayed directly through the HIP kernel here."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as LO
from oracle import msplat_oracle as MO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(H, W, seed, with_mask):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(H, W, 3, generator=g)
    rgb = (gt.permute(2, 0, 1) + 0.08 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    gtd = 1.0 + 3.0 * torch.rand(H, W, 1, generator=g)
    dm = (gtd.permute(2, 0, 1) * (1 + 0.1 * torch.randn(1, H, W, generator=g))).clamp(min=0.1)
    mask = None
    if with_mask:
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        mask = ((yy - H * 0.4) ** 2 + (xx - W * 0.6) ** 2) < (0.2 * H) ** 2
    return rgb, dm, gt, gtd, mask


@pytest.mark.parametrize("H,W,with_mask", [(48, 70, False), (48, 70, True), (100, 133, True)])
def test_image_loss_matches_oracle(H, W, with_mask):
    from gflow_amd.losses import image_loss
    rgb, dm, gt, gtd, mask = _inputs(H, W, 3, with_mask)
    lam_rgb, lam_d = 1.0, 0.1
    # oracle
    rc = rgb.clone().requires_grad_(True)
    dc = dm.clone().requires_grad_(True)
    ab_c = torch.tensor([1.1, -0.05], requires_grad=True)
    l_rgb, err_c = LO.rgb_loss(rc, gt, mask)
    l_dep = LO.depth_loss(dc, gtd, ab_c[0], ab_c[1], mask)
    (lam_rgb * l_rgb + lam_d * l_dep).backward()
    # HIP
    r4 = torch.cat([rgb, dm]).to(DEV).requires_grad_(True)
    ab_g = torch.tensor([1.1, -0.05], device=DEV, requires_grad=True)
    loss, err_g, lr_g, ld_g = image_loss(r4, gt.to(DEV), gtd.to(DEV), ab_g, lam_rgb, lam_d,
                                         None if mask is None else mask.to(DEV))
    loss.backward()
    assert abs(lr_g.item() - l_rgb.item()) <= 2e-5 * abs(l_rgb.item())
    assert abs(ld_g.item() - l_dep.item()) <= 2e-5 * abs(l_dep.item())
    np.testing.assert_allclose(err_g.cpu().numpy(), err_c.detach().numpy(), rtol=1e-5, atol=1e-8)
    g_ref = torch.cat([rc.grad, dc.grad])
    scale = g_ref.abs().max().item()
    np.testing.assert_allclose(r4.grad.cpu().numpy(), g_ref.numpy(), rtol=2e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(ab_g.grad.cpu().numpy(), ab_c.grad.numpy(), rtol=2e-4)


def test_ssim_golden_through_hip(golden_dir):
    from gflow_amd.losses import image_loss
    g = np.load(os.path.join(golden_dir, "ssim_small.npz"))
    for a, b, val, grad in (("img1", "img2", "value", "grad1"), ("img3", "img4", "value34", "grad3")):
        x = torch.from_numpy(g[a])[0]
        y = torch.from_numpy(g[b])[0]
        H, W = x.shape[1:]
        r4 = torch.cat([x, torch.zeros(1, H, W)]).to(DEV).requires_grad_(True)
        loss, _, loss_rgb, _ = image_loss(r4, y.permute(1, 2, 0).contiguous().to(DEV), None, None, 1.0, 0.0)
        loss.backward()
        mse = ((x - y) ** 2).mean()
        ssim_val = 1.0 - (loss_rgb.item() - mse.item())
        assert abs(ssim_val - float(g[val])) < 5e-6
        # d loss / dx = d mse/dx - d ssim/dx
        d_ssim = (2 * (x - y) / x.numel()) - r4.grad[:3].cpu()
        np.testing.assert_allclose(d_ssim.numpy(), g[grad][0], rtol=2e-3, atol=2e-8)


def test_ssim_480p_golden_probes(golden_dir):
    from gflow_amd.losses import image_loss
    g = np.load(os.path.join(golden_dir, "ssim_480p.npz"))
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = torch.rand(1, 3, 480, 854, generator=gen)
    y = (x + 0.1 * torch.rand(1, 3, 480, 854, generator=gen)).clamp(0, 1)
    r4 = torch.cat([x[0], torch.zeros(1, 480, 854)]).to(DEV).requires_grad_(True)
    loss, _, loss_rgb, _ = image_loss(r4, y[0].permute(1, 2, 0).contiguous().to(DEV), None, None, 1.0, 0.0)
    loss.backward()
    mse = ((x - y) ** 2).mean().item()
    assert abs((1.0 - (loss_rgb.item() - mse)) - float(g["value"])) < 5e-6
    for (c, i, j), ref in zip(g["probes"], g["grad_probes"]):
        d_mse = 2 * (x[0, c, i, j] - y[0, c, i, j]).item() / x.numel()
        got = d_mse - r4.grad[c, i, j].item()
        assert abs(got - ref) <= 2e-3 * abs(ref) + 1e-11


def test_adam_matches_torch_adam():
    from gflow_amd.optim import Adam, LinearLR
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(1000, 3, generator=g)
    q0 = torch.randn(7, generator=g)
    grads = [(torch.randn(1000, 3, generator=g), torch.randn(7, generator=g)) for _ in range(12)]
    pc, qc = p0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
    ref = torch.optim.Adam([{"params": [pc], "lr": 4e-3}, {"params": [qc], "lr": 1e-3}])
    sched = torch.optim.lr_scheduler.LinearLR(ref, start_factor=1.0, end_factor=0.1, total_iters=10)
    pg, qg = p0.clone().to(DEV).requires_grad_(True), q0.clone().to(DEV).requires_grad_(True)
    opt = Adam([{"params": [pg], "lr": 4e-3}, {"params": [qg], "lr": 1e-3}])
    sch = LinearLR(opt, start_factor=1.0, end_factor=0.1, total_iters=10)
    for gp, gq in grads:
        pc.grad, qc.grad = gp.clone(), gq.clone()
        ref.step(); sched.step()
        pg.grad, qg.grad = gp.to(DEV), gq.to(DEV)
        opt.step(); sch.step()
    np.testing.assert_allclose(pg.detach().cpu().numpy(), pc.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(qg.detach().cpu().numpy(), qc.detach().numpy(), rtol=2e-5, atol=2e-6)


def test_adam_row_mask_freezes_gradient_rows():
    from gflow_amd.optim import Adam
    p = torch.ones(8, 3, device=DEV, requires_grad=True)
    opt = Adam([p], lr=0.1)
    mask = torch.tensor([1, 0, 0, 1, 0, 0, 0, 1], dtype=torch.bool, device=DEV)
    opt.set_row_zero_grad(p, mask)
    p.grad = torch.ones_like(p)
    opt.step()
    out = p.detach().cpu()
    assert torch.all(out[mask.cpu()] == 1.0)           # zero grad, zero moments -> no move
    assert torch.all(out[~mask.cpu()] < 1.0)


def test_colormap_matches_golden(golden_dir):
    from gflow_amd.color import apply_float_colormap, lut
    g = np.load(os.path.join(golden_dir, "colormap.npz"))
    out = apply_float_colormap(torch.from_numpy(g["depth_vec"]).to(DEV), "turbo", non_zero=True)
    np.testing.assert_array_equal(out.cpu().numpy(), g["turbo_non_zero"])
    rb = apply_float_colormap(torch.from_numpy(g["ramp"]).to(DEV), "gist_rainbow")
    np.testing.assert_array_equal(rb.cpu().numpy(), g["rainbow_ramp"])
    # random depth vector incl. zeros vs the oracle restatement
    gen = torch.Generator().manual_seed(4)
    d = 0.5 + 4 * torch.rand(5000, 1, generator=gen)
    d[::17] = 0
    ref = MO.apply_float_colormap(d, MO.turbo_lut(), non_zero=True)
    got = apply_float_colormap(d.to(DEV), "turbo", non_zero=True).cpu()
    mism = (ref != got).any(dim=1).double().mean().item()
    assert mism <= 2e-3            # (x*255).long() may land one bin over on a rounding edge


@pytest.mark.parametrize("with_mask", [False, True])
def test_cached_target_statistics_match_full_kernel(with_mask):
    """gfl_loss_prepare_gt + gfl_loss_fwd_bwd_partials_cached (three filtered maps, the target's two
    read back) against gfl_loss_fwd_bwd_partials (five maps): the same numbers up to the compiler's
    choice of fused multiply-adds in the two instantiations."""
    import ctypes
    from gflow_amd import _lib as L
    lib = L.load()
    H, W = 70, 101
    g = torch.Generator().manual_seed(11)
    render = torch.rand(4, H, W, generator=g).to(DEV)
    gt = torch.rand(H, W, 3, generator=g).to(DEV)
    gd = (1 + torch.rand(H, W, generator=g)).to(DEV)
    keep = (torch.rand(H, W, generator=g) > 0.2).to(torch.uint8).to(DEV) if with_mask else None
    ab = torch.tensor([1.1, -0.05], device=DEV)

    def run(cached):
        d = torch.empty_like(render)
        err = torch.empty(H, W, device=DEV)
        ws = torch.zeros(lib.gfl_loss_workspace_bytes(W, H), dtype=torch.uint8, device=DEV)
        ps, pg = ctypes.c_void_p(), ctypes.c_void_p()
        ns, ng = ctypes.c_int(), ctypes.c_int()
        common = (L.ptr(render), L.ptr(gt), L.ptr(gd), L.ptr(keep), L.ptr(ab), 1.0, 0.1, W, H, L.ptr(d), L.ptr(err),
                  L.ptr(ws), ws.numel())
        tail = (ctypes.byref(ps), ctypes.byref(ns), ctypes.byref(pg), ctypes.byref(ng), L.stream())
        if cached:
            stats = torch.empty(6, H, W, device=DEV)
            L.check(lib.gfl_loss_prepare_gt(L.ptr(gt), L.ptr(keep), W, H, L.ptr(stats), L.stream()), "prepare")
            L.check(lib.gfl_loss_fwd_bwd_partials_cached(*common, L.ptr(stats), *tail), "cached")
        else:
            L.check(lib.gfl_loss_fwd_bwd_partials(*common, *tail), "plain")
        torch.cuda.synchronize()
        n = ns.value
        off = ps.value - ws.data_ptr()
        part = ws[off:off + 4 * n].view(torch.float32).clone()
        return d, err, part

    d0, e0, p0 = run(False)
    d1, e1, p1 = run(True)
    assert torch.equal(e0, e1)
    scale = d0.abs().max().item()
    assert (d0 - d1).abs().max().item() <= 2e-6 * scale
    torch.testing.assert_close(p0, p1, rtol=1e-6, atol=1e-5)
