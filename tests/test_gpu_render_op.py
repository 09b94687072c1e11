"""The fused rasteriser as ONE differentiable operator -- gflow_amd.render.render(gaussians, camera), i.e.
gfl_render_fwd / gfl_render_bwd of include/gflow_hip.h -- against the CPU oracle's render_multiple and against the
operator-by-operator HIP path (``-m gpu``).  This is what a GFlow user gets after the one-line ``import msplat`` swap:
gflow/utils/render.py:6-108 asks for exactly {"rgb", "uv", "depth", "depth_map"} in the training call."""
import pytest
import torch

from oracle import msplat_oracle as MO
from tests.scenes import random_scene, scene_group
from tests.test_gpu_parity import close_frac, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ("xyz", "scale", "rotate", "opacity", "rgb")


def _weights(H, W, n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g), 1e-2 * torch.randn(n, 2, generator=g),
            1e-1 * torch.randn(n, 1, generator=g))


def _loss(out, w):
    return (out["rgb"] * w[0]).sum() + (out["depth_map"] * w[1]).sum() + (out["uv"] * w[2]).sum() + (out["depth"] * w[3]).sum()


@pytest.mark.parametrize("bg", [0.0, 0.33])
def test_render_operator_values_and_gradients_match_oracle(bg):
    import gflow_amd.render as R
    s = random_scene(3000, 200, 136, seed=11, sigma_px=2.5)
    n, W, H = s["xyz"].shape[0], s["W"], s["H"]
    w = _weights(H, W, n, 5)
    # oracle
    leaves_c = {k: s[k].clone().requires_grad_(True) for k in NAMES}
    extr_c = s["extr"].clone().requires_grad_(True)
    oc = MO.render_multiple([*[leaves_c[k] for k in NAMES], s["intr"], extr_c, bg, W, H], ["rgb", "uv", "depth", "depth_map"])
    _loss(oc, w).backward()
    # fused operator
    leaves_g = {k: s[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
    extr_g = s["extr"].clone().to(DEV).requires_grad_(True)
    og = R.render(leaves_g, dict(intr=s["intr"].to(DEV), extr=extr_g, W=W, H=H), bg)
    for k in ("rgb", "depth_map"):
        close_frac(og[k], oc[k], 1e-4, 1e-5, bad_frac=3e-4, hard=2e-2, what=k)
    close_frac(og["uv"], oc["uv"], 1e-5, 1e-3, what="uv")
    close_frac(og["depth"], oc["depth"], 1e-6, 1e-6, what="depth")
    _loss(og, [t.to(DEV) for t in w]).backward()
    for k in NAMES:
        ref = leaves_c[k].grad
        got = leaves_g[k].grad.cpu()
        rel = (got - ref).norm() / ref.norm()
        assert rel < 2e-3, f"d_{k}: relative L2 error {rel:.2e}"
    rel = (extr_g.grad.cpu() - extr_c.grad).norm() / extr_c.grad.norm()
    assert rel < 2e-3, f"d_extr: relative L2 error {rel:.2e}"


def test_render_multiple_routes_the_training_call_through_the_fused_operator():
    """Same numbers from render_multiple whether it composes the five msplat operators or calls the fused one; and
    the fused one is really what runs for the training call's output set."""
    import gflow_amd.render as R
    s = random_scene(2500, 168, 120, seed=3, sigma_px=2.0)
    d = to_dev(s)
    want = ["rgb", "uv", "depth", "depth_map"]
    calls = []
    orig = R._FusedRender.apply
    R._FusedRender.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        fused = R.render_multiple(scene_group(d, 0.0), want)
        assert calls, "render_multiple did not use the fused operator"
        calls.clear()
        R.render_multiple(scene_group(d, 0.0), want + ["center"])          # snapshot images: operator path
        assert not calls
    finally:
        R._FusedRender.apply = orig
    R.USE_FUSED = False
    try:
        ops = R.render_multiple(scene_group(d, 0.0), want)
    finally:
        R.USE_FUSED = True
    close_frac(fused["rgb"], ops["rgb"], 2e-5, 2e-6, bad_frac=1e-4, hard=2e-2, what="rgb fused vs operators")
    close_frac(fused["depth_map"], ops["depth_map"], 2e-5, 2e-6, bad_frac=1e-4, hard=2e-2, what="depth_map")
    close_frac(fused["uv"], ops["uv"], 1e-6, 1e-4, what="uv")            # (same formulas, separately compiled kernels)
    close_frac(fused["depth"], ops["depth"], 1e-6, 1e-6, what="depth")


def test_two_forwards_before_their_backwards_and_empty_input():
    """An engine stays with its graph from forward to backward: a second forward in between gets another one, and
    both backwards give what they give alone.  N = 0 (a boolean-mask gather that selected nothing,
    trainer.py:430,658) renders the background."""
    import gflow_amd.render as R
    sa = to_dev(random_scene(1200, 96, 80, seed=1, sigma_px=2.0))
    sb = to_dev(random_scene(900, 96, 80, seed=2, sigma_px=3.0))

    def run(s, together_with=None):
        leaves = {k: s[k].clone().requires_grad_(True) for k in NAMES}
        out = R.render(leaves, dict(intr=s["intr"], extr=s["extr"], W=s["W"], H=s["H"]), 0.0)
        other = together_with() if together_with else None
        (out["rgb"].sum() + out["depth_map"].sum()).backward()
        return {k: leaves[k].grad.clone() for k in NAMES}, other

    alone_a, _ = run(sa)
    alone_b, _ = run(sb)
    nested_a, nested_b = run(sa, together_with=lambda: run(sb)[0])
    for k in NAMES:
        # (two runs of the same backward differ by the order of its LDS adds: ~1e-8 of the largest entry, which is more than
        #  1e-4 of an entry that is itself a difference of large terms -- the absolute floor follows the tensor's scale)
        for x, y in ((alone_a[k], nested_a[k]), (alone_b[k], nested_b[k])):
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()) + 1e-7), k
    # N = 0
    e = {k: sa[k][:0] for k in NAMES}
    out = R.render(e, dict(intr=sa["intr"], extr=sa["extr"], W=96, H=80), 0.33)
    assert out["rgb"].shape == (3, 80, 96) and torch.allclose(out["rgb"], torch.full_like(out["rgb"], 0.33))
    assert out["uv"].shape == (0, 2) and out["depth"].shape == (0, 1)


def test_a_stale_finalizer_cannot_free_an_engine_that_is_reserved_again():
    """backward() used to free the pooled engine while the finalizer registered in forward stayed armed; it
    fired when the OLD graph node was collected -- after the next forward had taken the same engine -- and a third
    render then overwrote the state of a forward / backward pair that was still open.  Reservations are tokens now."""
    import gc
    import gflow_amd.render as R
    sa = to_dev(random_scene(1000, 96, 80, seed=5, sigma_px=2.0))
    sb = to_dev(random_scene(700, 96, 80, seed=6, sigma_px=3.0))
    cam = lambda s: dict(intr=s["intr"], extr=s["extr"], W=s["W"], H=s["H"])

    def fwd(s):
        leaves = {k: s[k].clone().requires_grad_(True) for k in NAMES}
        return leaves, R.render(leaves, cam(s), 0.0)

    la, oa = fwd(sa)
    (oa["rgb"].sum() + oa["depth_map"].sum()).backward()
    alone = {k: la[k].grad.clone() for k in NAMES}
    pool = R._POOL[(96, 80, str(sa["xyz"].device))]
    n_busy = lambda: sum(1 for e in pool if e.busy)
    assert n_busy() == 0
    old_leaves, old_out = fwd(sb)                                   # iteration 1 of "b": forward + backward
    (old_out["rgb"].sum()).backward()
    assert n_busy() == 0
    la2, oa2 = fwd(sa)                                              # takes the engine that "b" just gave back
    assert n_busy() == 1
    del old_leaves, old_out                                         # the old graph node goes away NOW
    gc.collect()
    assert n_busy() == 1, "a stale finalizer freed an engine that another forward holds"
    lb, ob = fwd(sb)                                                # must not get (and overwrite) that engine
    assert n_busy() == 2
    (ob["rgb"].sum()).backward()
    (oa2["rgb"].sum() + oa2["depth_map"].sum()).backward()
    assert n_busy() == 0
    for k in NAMES:
        assert torch.allclose(alone[k], la2[k].grad, rtol=1e-4, atol=1e-7), k
    # a second backward through the same forward has no engine any more: a clear error, not a wrong gradient
    lc, oc = fwd(sa)
    loss = oc["rgb"].sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
    # a forward that is dropped without a backward gives its engine back
    ld, od = fwd(sa)
    assert n_busy() == 1
    del ld, od
    gc.collect()
    assert n_busy() == 0


def test_operator_cost_is_reported_not_asserted():
    """what the three levels of the drop-in cost at 480p / 60k -- the five msplat operators one by one
    (the one-line ``import msplat`` swap), the fused ``render`` operator, the whole fused fit iteration.  REPORTED (bench.py
    carries the same figures in its line as ``drop_in_levels``), not asserted: half of the operator's time is host time, and a
    wall-clock ratio has no place under ``pytest -x`` beside parity (the old guard was moved 2.0x -> 3.5x after
    it failed one run in twenty-five).  What IS held: all three levels run and the two operator levels give the same image."""
    from bench import drop_in_levels
    out = drop_in_levels(torch.device(DEV), 480, 854, 60000, repeats=2, n=10)
    print("drop-in levels (ms per forward + backward):", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()})
    assert out["five_operators_ms"] > 0 and out["fused_render_ms"] > 0 and out["fit_iteration_ms"] > 0
    assert out["rgb_max_abs_diff"] < 1e-4
