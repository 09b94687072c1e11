"""BASELINE.json configs[0] on the HIP path (``-m gpu``): one 480x854 frame, 10 000 splats, 200 first-frame iterations --
the fused iteration (gfl_fit_iterations through FitEngine, hipGraph replay as a fit runs it) against the eager-CPU fit
(oracle.fit_oracle.OracleFit: torch.optim.Adam + LinearLR over the oracle's operators, gflow/trainer.py:387-558) run for the
SAME 200 iterations from the same image-driven initial splats (trainer.py:206-238; lr 4e-3 as scripts/fit_video.sh:16-39,
lambda rgb / depth / var = 1 / 0.1 / 10 as the bench).

A 200-step Adam trajectory is chaotic in its low bits (an update is +-lr whatever the gradient's size), so what is held is
what a user of the fit sees: the two loss terms at every 20th iteration, the final PSNR, and -- while the trajectories are
still the same trajectory -- the rows themselves after 10 steps.  Observed figures are printed.
"""
import pytest
import torch

from oracle import fit_oracle as FO
from tests.test_gpu_fused import _engine

pytestmark = pytest.mark.gpu
H, W, N, ITERS = 480, 854, 10000, 200
NAMES = ("xyz", "scale", "rotate", "opacity", "rgb")


def _psnr(rgb, target):          # rgb (3, H, W), target (H, W, 3)
    return float(10.0 * torch.log10(1.0 / ((rgb.clamp(0, 1) - target.permute(2, 0, 1)) ** 2).mean()))


def test_config0_10k_splats_200_iterations_track_the_oracle_fit():
    from gflow_amd import synthetic as S
    from gflow_amd.fused import COLS
    frame = S.make_frame(H, W, seed=0)
    raw = S.init_splats(frame, N, seed=0)
    lr = 4e-3
    lam = dict(lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0)
    eng = _engine({k: raw[k] for k in NAMES}, dict(W=W, H=H, intr=raw["intr"]), frame["image"], frame["depth"], lr=lr,
                  lr_camera=0.0, total_iters=ITERS, **lam)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    fit = FO.OracleFit(raw, raw["intr"], frame, lr=lr, iterations=ITERS, **lam)
    curve, checks = [], []
    for it in range(ITERS):
        # (the HIP fit through hipGraph replays, as a fit runs it)
        looked_at = it % 20 == 0 or it == ITERS - 1 or it == 9
        eng.iteration(use_graph=it > 0)
        _, info = fit.step()
        if it == 9:
            # Ten steps in.  An Adam step is +-lr whatever the gradient's size, so an entry whose gradient is at rounding level
            # is a coin flip in BOTH fits -- most entries here: the image-driven initial splats are a fraction of a pixel wide
            # (trainer.py:206-238), their rotation does not enter the image at all, the three equal scales sit on the kink of the
            # var term -- and the entries that move are set on their course by the first of those flips.  Held: the attributes
            # with a definite gradient (colour, opacity), on the entries the oracle's fit moved steadily (>= 8 lr in ten steps);
            # the rest is printed.
            for k, (a, b) in COLS.items():
                ref = fit.raw[k].detach().reshape(N, b - a)
                moved = (ref - raw[k].reshape(N, b - a)).abs() >= 8.0 * lr
                d = (eng.params[:N, a:b].cpu() - ref).abs()
                off = (d[moved] > lr).double().mean().item() if bool(moved.any()) else 0.0
                print(f"observed config0 step 10: {k}: {moved.double().mean().item():.3f} of the entries moved steadily; of those "
                      f"{off:.4f} are off by more than lr (all entries: {(d > lr).double().mean().item():.4f})")
                checks.append((k, float(moved.double().mean()), off))
            # ... and what the rows ARE for: the two fits' renders of the tenth iteration
            mse = ((eng.render.cpu() - info["render4"].detach()) ** 2)[:3].mean().item()
            psnr10 = -10.0 * float(torch.log10(torch.tensor(mse)))
            print(f"observed config0 step 10: HIP render against the oracle's {psnr10:.2f} dB")
        if looked_at:
            l_rgb, l_depth = (float(x) for x in eng.loss_terms())
            curve.append((it, l_rgb, float(info["l_rgb"]), l_depth, float(info["l_depth"])))
    eng.check_overflow()
    assert int(eng.step.item()) == ITERS
    for it, a, b, c, d in curve:
        print(f"observed config0 it {it:3d}: l_rgb {a:.5f} / {b:.5f} ({abs(a - b) / b:.2e})  l_depth {c:.5f} / {d:.5f} ({abs(c - d) / d:.2e})")
    for k, share, off in checks:
        if k in ("rgb", "opacity"):
            assert share > 0.03 and off < 0.1, f"{k}: {off:.3f} of the steadily moving entries are off by more than lr after ten steps"
    # (lr 4e-3 in xyz is a pixel per step at this scene's depth: the coin flips above ARE visible, pixel by pixel -- observed
    #  31.6 dB -- while the loss terms agree to 0.5 % at the same iteration)
    assert psnr10 > 28.0, psnr10
    for it, a, b, c, d in curve:
        # the loss the fit minimises (the var term is the same function of the rows in both) and its two image terms
        ta, tb = a + 0.1 * c, b + 0.1 * d
        assert abs(ta - tb) <= 0.02 * tb, f"iteration {it}: loss {ta} against the oracle's {tb}"
        assert abs(a - b) <= 0.02 * b, f"iteration {it}: l_rgb {a} against the oracle's {b}"
        assert abs(c - d) <= 0.05 * d, f"iteration {it}: l_depth {c} against the oracle's {d}"      # (a tenth of the loss's weight)
    assert curve[-1][1] < 0.5 * curve[0][1]                                  # ... and it is a fit: the loss halves
    # final PSNR of the two fits' renders (one more forward each, on the stepped rows)
    eng.forward()
    with torch.no_grad():
        _, info = FO.fit_loss(fit.raw, fit.pose, fit.depth_ab, raw["intr"], frame, 0.0, 1.0, 0.1, 10.0)
    p_hip, p_cpu = _psnr(eng.render[:3].cpu(), frame["image"]), _psnr(info["render4"][:3], frame["image"])
    print(f"observed config0: PSNR after {ITERS} iterations {p_hip:.3f} dB (HIP) / {p_cpu:.3f} dB (CPU oracle)")
    assert abs(p_hip - p_cpu) < 0.2, (p_hip, p_cpu)
