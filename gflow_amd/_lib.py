"""ctypes binding of libgflow_hip.so (the C ABI declared in include/gflow_hip.h).

There is NO CPU fallback: if the library is missing or a tensor is not on a HIP
device the call raises.  The library is built in-tree by ``build()`` (hipcc
cross-compiles gfx950 without a GPU) and travels with the repository snapshot.
"""
import ctypes
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# GFLOW_HIP_LIB: another build of the same sources (e.g. one made with other rasteriser constants,
# make CONSTS="-DGFL_PIXEL_CENTER=0.5f": include/gflow_hip.h, tests/test_gpu_pixel_center.py).  Still the HIP library -- there
# is no other kind.
LIB_PATH = os.environ.get("GFLOW_HIP_LIB") or os.path.join(_HERE, "libgflow_hip.so")

_lock = threading.Lock()
_lib = None

c_void_p, c_int, c_float, c_size_t, c_int64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64

MIN_VERSION = 304          # GFL_VERSION of the include/gflow_hip.h this binding mirrors (overflow[4], sort-order trailer, gfl_constants_n)

# name -> (restype, argtypes); mirrors include/gflow_hip.h one to one
_P = c_void_p
SIGNATURES = {
    "gfl_version": (c_int, []),
    "gfl_constants": (c_int, [_P]),
    "gfl_constants_n": (c_int, [_P, c_int]),
    "gfl_ewa_on_mfma": (c_int, []),
    "gfl_status_string": (ctypes.c_char_p, [c_int]),
    "gfl_last_hip_error": (c_int, []),
    "gfl_reduce_workspace_bytes": (c_size_t, [c_int]),
    "gfl_project_point_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P]),
    "gfl_project_point_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_size_t, _P]),
    "gfl_cov3d_fwd": (c_int, [_P, _P, _P, c_int, _P, _P]),
    "gfl_cov3d_bwd": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P]),
    "gfl_ewa_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "gfl_ewa_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "gfl_bin_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gfl_bin_count": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "gfl_bin_sort": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "gfl_blend_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_float, c_int, c_int, _P, _P, _P, _P]),
    "gfl_blend_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_float, c_int, c_int, _P, _P, _P, c_int,
                              _P, _P, _P, _P, c_int, _P]),
    "gfl_sh_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "gfl_sh_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    "gfl_colormap_nonzero": (c_int, [_P, c_int, _P, _P, _P, c_size_t, _P]),
    "gfl_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gfl_loss_fwd_bwd": (c_int, [_P, _P, _P, _P, _P, c_float, c_float, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "gfl_loss_fwd_bwd_partials": (c_int, [_P, _P, _P, _P, _P, c_float, c_float, c_int, c_int, _P, _P, _P, c_size_t, _P, _P, _P, _P, _P]),
    "gfl_loss_prepare_gt": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "gfl_loss_fwd_bwd_partials_cached": (c_int, [_P, _P, _P, _P, _P, c_float, c_float, c_int, c_int, _P, _P, _P, c_size_t, _P, _P, _P, _P, _P, _P]),
    "gfl_adam_step": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P, c_float, c_float, c_float, c_float, _P, c_float, c_int, _P]),
    "gfl_step_increment": (c_int, [_P, _P]),
    "gfl_tile_sort_only": (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
    "gfl_tile_sort_ordered": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "gfl_tile_sort_reserved": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "gfl_fit_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gfl_fit_forward": (c_int, [_P, _P, _P]),          # struct pointers; typed in gflow_amd/fused.py
    "gfl_fit_backward_step": (c_int, [_P, _P, _P]),
    "gfl_fit_iteration": (c_int, [_P, _P, _P]),
    "gfl_fit_iterations": (c_int, [_P, _P, c_int, c_int, _P]),
    "gfl_fit_reserved_supported": (c_int, [_P, _P]),
    "gfl_fit_snapshot_stage": (c_int, [_P, _P, _P]),
    "gfl_fit_snapshot_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gfl_fit_snapshot": (c_int, [_P, _P, _P, _P, _P, c_size_t, _P]),
    "gfl_fit_iteration_snapshot": (c_int, [_P, _P, _P, _P, _P]),
    "gfl_render_fwd": (c_int, [_P, _P, _P]),
    "gfl_render_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "gfl_fit_prepare_targets": (c_int, [_P, _P]),
    "gfl_fit_schedule_info": (c_int, [_P, _P, _P, _P, _P]),
    "gfl_fit_schedule_info_fwd": (c_int, [_P, _P, _P, _P, _P]),
    "gfl_selftest_reduce10": (c_int, [_P, _P, _P, _P]),
    "gfl_selftest_cov2d": (c_int, [_P, _P, c_int, _P, _P, _P]),
    "gfl_selftest_block_mask": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "gfl_concave_hull": (c_int, [_P, c_int, ctypes.c_double, ctypes.c_double, _P, c_int]),
    "gfl_abi_sizes": (c_int, [_P, _P]),
    "gfl_profile_enable": (c_int, [ctypes.c_uint]),
    "gfl_profile_read": (c_int, [_P, _P, c_int]),
}


def build(force=False, quiet=True):
    """Compile libgflow_hip.so for gfx950 with hipcc (in-tree, via the Makefile)."""
    if force:
        subprocess.run(["make", "-C", _CSRC, "clean"], check=True, capture_output=quiet)
    res = subprocess.run(["make", "-C", _CSRC, "-j8"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libgflow_hip.so failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


def load():
    """dlopen the library and attach prototypes; raises if it is not there."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C gflow_amd/csrc`).  gflow_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the ABI drifted
            fn.restype = res
            fn.argtypes = args
        if lib.gfl_version() < MIN_VERSION:
            raise RuntimeError(f"{LIB_PATH} is version {lib.gfl_version()}, this binding needs >= {MIN_VERSION}: rebuild it "
                               "(make -C gflow_amd/csrc)")
        _lib = lib
        return lib


def check(rc, what=""):
    if rc != 0:
        lib = load()
        msg = lib.gfl_status_string(rc).decode()
        extra = f" (hipError {lib.gfl_last_hip_error()})" if rc == -3 else ""
        raise RuntimeError(f"libgflow_hip: {what}: {msg}{extra}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def need_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gflow_amd: tensors must be on a HIP (cuda) device; there is no CPU fallback")


def scratch(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------- CU-masked streams
_hip = None


def _hip_runtime():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")         # (already in the process: torch links it)
        _hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
        _hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32,
                                                      ctypes.POINTER(ctypes.c_uint32)]
    return _hip


def cu_partition(parts, device=None, cus=None):
    """Split the device's compute units into ``parts`` disjoint shares that each span ALL XCDs: share p gets the CUs
    [p * n, (p + 1) * n) of every XCD, n = (CUs per XCD) // parts.  Returns [(mask words, CU count)] per share.
    Mask layout on gfx950 (tools/cumask_probe.hip, measured): bit i of the mask = CU i // 8 of XCD i % 8 -- a mask that leaves
    an XCD without a CU is not honoured at all (the stream then runs on the whole device), so a share cannot be 'four of the
    eight XCDs'; what can be split is every XCD's 32 CUs.  Launches AND graph launches on a masked stream stay inside its
    mask; two disjoint shares run side by side without slowing each other (the probe's spin kernel: 52.8 ms on one half
    alone, 53.3 ms on both halves at once, 29.0 ms on the whole chip)."""
    if cus is None:
        cus = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
    xcds = 8
    per_xcd = cus // xcds
    n = per_xcd // int(parts)
    if n < 1:
        raise ValueError(f"cu_partition: {parts} shares of {per_xcd} CUs per XCD")
    out = []
    for p in range(int(parts)):
        words = [0] * ((cus + 31) // 32)
        for cu in range(p * n, (p + 1) * n):
            for x in range(xcds):
                i = cu * xcds + x
                words[i // 32] |= 1 << (i % 32)
        out.append((words, n * xcds))
    return out


def masked_stream(words, device=None):
    """A torch stream whose kernels run only on the compute units of ``words`` (hipExtStreamCreateWithCUMask), as a
    torch.cuda.ExternalStream.  The HIP stream lives until the process ends (a handful per process)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        arr = (ctypes.c_uint32 * len(words))(*words)
        h = ctypes.c_void_p()
        rc = _hip_runtime().hipExtStreamCreateWithCUMask(ctypes.byref(h), len(words), arr)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: hipError {rc}")
    return torch.cuda.ExternalStream(h.value, device=dev)
