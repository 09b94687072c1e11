"""Device self-tests of wave-level primitives (``-m gpu``)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_wave_reduce_scatter10_sums_every_component():
    from gflow_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 10, generator=g)
    # make every (lane, component) entry identifiable: integer weights catch any lane/slot mix-up
    x = x + torch.arange(10).float() * 100.0
    xd = x.cuda().contiguous()
    a = torch.full((10,), float("nan"), device="cuda")
    b = torch.full((10,), float("nan"), device="cuda")
    L.check(lib.gfl_selftest_reduce10(L.ptr(xd), L.ptr(a), L.ptr(b), L.stream()), "selftest")
    ref = x.double().sum(0)
    assert torch.allclose(a.cpu().double(), ref, rtol=1e-5, atol=1e-3), (a.cpu(), ref)
    assert torch.allclose(b.cpu().double(), ref, rtol=1e-5, atol=1e-3), (b.cpu(), ref)


def test_ewa_contraction_on_the_matrix_cores_equals_the_valu_form():
    """Sigma2 = M Sigma M^T with v_mfma_f32_4x4x1_16b_f32 (sixteen splats per instruction, quad broadcasts to lay
    them out) against the 30-FMA form and against float64, over magnitudes the preprocess kernel sees."""
    from gflow_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    n = 1000                                                  # not a multiple of 64: partial waves pass zeros
    m = (torch.randn(n, 6, generator=g) * 300.0).contiguous()
    r = torch.randn(n, 3, 3, generator=g) * torch.logspace(-3, 0, n).reshape(n, 1, 1)
    S = r @ r.transpose(1, 2)                                 # symmetric positive semi-definite
    cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1).contiguous()
    a = torch.full((n, 3), float("nan"), device="cuda")
    b = torch.full((n, 3), float("nan"), device="cuda")
    md, cd = m.cuda(), cov.cuda()
    L.check(lib.gfl_selftest_cov2d(L.ptr(md), L.ptr(cd), n, L.ptr(a), L.ptr(b), L.stream()), "selftest")
    torch.cuda.synchronize()
    M = m.reshape(n, 2, 3).double()
    ref = M @ S.double() @ M.transpose(1, 2)
    ref = torch.stack([ref[:, 0, 0], ref[:, 0, 1], ref[:, 1, 1]], dim=1)
    scale = ref.abs().max(dim=1, keepdim=True).values
    assert ((a.cpu().double() - ref).abs() / scale).max() < 1e-5
    assert ((b.cpu().double() - ref).abs() / scale).max() < 1e-5
    assert ((a - b).abs().cpu().double() / scale).max() < 1e-6
