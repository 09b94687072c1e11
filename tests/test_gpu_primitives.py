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
