"""CPU: the C-ABI library builds, loads and exports every symbol include/gflow_hip.h declares,
the ctypes table mirrors the header, and the product path refuses to run without a device."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "gflow_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gflow_amd import _lib
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gflow_hip.h but not exported"


def test_ctypes_table_covers_the_header():
    from gflow_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "gflow_hip.h")).read()
    assert lib.gfl_version() == int(re.search(r"^#define GFL_VERSION (\d+)$", hdr, flags=re.M).group(1)) >= _lib.MIN_VERSION
    assert lib.gfl_status_string(-2) == b"workspace too small"


def test_struct_mirrors_match_the_c_layout():
    from gflow_amd import _lib, fused
    lib = _lib.load()
    a, b = ctypes.c_int(), ctypes.c_int()
    assert lib.gfl_abi_sizes(ctypes.byref(a), ctypes.byref(b)) == 0
    assert a.value == ctypes.sizeof(fused.FitState)
    assert b.value == ctypes.sizeof(fused.FitHyper)


def test_size_queries_without_a_gpu():
    from gflow_amd import _lib
    lib = _lib.load()
    assert lib.gfl_reduce_workspace_bytes(60000) == 235 * 12 * 4
    assert lib.gfl_bin_workspace_bytes(60000, 300000, 854, 480) > 300000 * 8
    assert lib.gfl_loss_workspace_bytes(854, 480) > 9 * 854 * 480 * 4
    assert lib.gfl_fit_workspace_bytes(60000, 1000000, 854, 480) > 1000000 * (8 + 48)
    # invalid arguments are rejected before any launch
    assert lib.gfl_project_point_fwd(None, None, None, 5, 8, 8, 0.2, 1.3, None, None, None) == -1
    assert lib.gfl_blend_fwd(None, None, None, None, 3, 0, 5, None, None, 0.0, 8, 8, None, None, None, None) == -1


def test_product_path_has_no_cpu_fallback():
    import gflow_amd.msplat as ms
    from gflow_amd.trainer import SimpleGaussian
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ms.project_point(torch.zeros(4, 3), torch.zeros(4), torch.zeros(3, 4), 8, 8)
    with pytest.raises(RuntimeError):
        SimpleGaussian(torch.zeros(8, 8, 3), torch.zeros(8, 8, 1), num_points=4, device="cpu")


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gflow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_library_and_oracle_use_the_same_constants():
    """The rasteriser constants are assumptions about msplat's internals, kept in ONE place per side:
    include/gflow_hip.h (overridable at build time) and the top of oracle/msplat_oracle.py.  They must agree."""
    import ctypes
    from gflow_amd import _lib
    from oracle import msplat_oracle as MO
    out = (ctypes.c_float * 10)()
    assert _lib.load().gfl_constants(out) == 0
    want = [MO.TILE, MO.NEAREST, MO.EXTENT, MO.FOV_CLAMP, MO.LOWPASS, MO.EIG_FLOOR, MO.RADIUS_SIGMA, MO.ALPHA_MIN,
            MO.ALPHA_MAX, MO.T_MIN]
    for got, ref in zip(out, want):
        assert abs(got - ref) <= 1e-7 * max(1.0, abs(ref)), (list(out), want)
    # the whole list, the tenth assumption of SURVEY.md 8c included: where a pixel is sampled (GFL_PIXEL_CENTER)
    hdr = open(os.path.join(ROOT, "include", "gflow_hip.h")).read()
    n_hdr = int(re.search(r"^#define GFL_N_CONSTANTS (\d+)$", hdr, flags=re.M).group(1))
    lib = _lib.load()
    lib.gfl_constants_n.restype = ctypes.c_int
    lib.gfl_constants_n.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    full = (ctypes.c_float * 16)(*([-7.0] * 16))
    assert lib.gfl_constants_n(full, 16) == n_hdr == len(want) + 1
    want_full = want + [MO.PIXEL_CENTER]
    for got, ref in zip(full, want_full):
        assert abs(got - ref) <= 1e-7 * max(1.0, abs(ref)), (list(full), want_full)
    assert full[n_hdr] == -7.0                       # nothing written past the list
    short = (ctypes.c_float * 4)(*([-7.0] * 4))
    assert lib.gfl_constants_n(short, 3) == n_hdr and abs(short[2] - want[2]) < 1e-6 and short[3] == -7.0


def test_iteration_flags_of_the_host_side_are_the_headers():
    """FitEngine passes GFL_ITER_RESERVED as a literal (the reserved tile regions of include/gflow_hip.h): it must be the
    header's value, and the flags must be distinct bits."""
    from gflow_amd.fused import FitEngine
    hdr = open(os.path.join(ROOT, "include", "gflow_hip.h")).read()
    flags = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define (GFL_ITER_[A-Z_]+) (\d+)$", hdr, flags=re.M)}
    assert set(flags) == {"GFL_ITER_RESERVED"}
    assert FitEngine.GFL_ITER_RESERVED == flags["GFL_ITER_RESERVED"]
    bits = sorted(flags.values())
    assert all(b & (b - 1) == 0 for b in bits) and len(set(bits)) == len(bits)


def test_reserved_entry_points_reject_bad_arguments_without_a_gpu():
    """gfl_tile_sort_reserved / gfl_fit_reserved_supported validate their arguments before anything is launched."""
    import ctypes
    from gflow_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.gfl_tile_sort_reserved(null, null, null, null, 64, 64, 16, null, null, null, null) != 0
    assert lib.gfl_fit_reserved_supported(null, null) == 0
