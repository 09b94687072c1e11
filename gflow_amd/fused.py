"""Fused fit engine: one library call per optimisation iteration.

``FitEngine`` owns the capacity-based HBM buffers the fused kernels work on (packed
64-byte parameter rows + Adam moments, per-splat render records, tile lists, image
planes) and drives ``gfl_fit_forward`` / ``gfl_fit_backward_step`` /
``gfl_fit_iteration`` (include/gflow_hip.h).  Nothing in an iteration touches the host:
no allocation, no read-back, launch sizes depend on N, the tile grid and the image only,
so a whole iteration can be captured in a hipGraph and replayed.
"""
import ctypes

import torch

from . import _lib as L

ROW = 16
REC = 12
COLS = {"xyz": (0, 3), "scale": (3, 6), "rotate": (6, 10), "opacity": (10, 11), "rgb": (11, 14)}

_P = ctypes.c_void_p
_I = ctypes.c_int32


class FitState(ctypes.Structure):          # mirrors gfl_fit_state
    _fields_ = [("N", _I), ("cap", _I), ("W", _I), ("H", _I), ("K_cap", _I), ("gt_cached", _I),
                ("params", _P), ("adam_m", _P), ("adam_v", _P), ("rec", _P), ("d_rec", _P),
                ("flow_target", _P), ("flow_w", _P), ("still_target", _P), ("still_w", _P), ("row_flags", _P),
                ("pose", _P), ("pose_m", _P), ("pose_v", _P),
                ("depth_ab", _P), ("depth_ab_m", _P), ("depth_ab_v", _P),
                ("intr", _P), ("extr", _P), ("d_extr", _P), ("step", _P),
                ("gt_rgb", _P), ("gt_depth", _P), ("keep", _P), ("move_mask", _P), ("foot_flags", _P),
                ("render", _P), ("final_T", _P), ("n_contrib", _P),
                ("d_render", _P), ("err_px", _P), ("sums", _P),
                ("tile_offsets", _P), ("ids", _P), ("tile_range", _P), ("overflow", _P),
                ("workspace", _P), ("workspace_bytes", ctypes.c_size_t), ("cu_count", _I), ("reserved_", _I)]


class FitHyper(ctypes.Structure):          # mirrors gfl_fit_hyper
    _fields_ = [("bg", ctypes.c_float), ("nearest", ctypes.c_float), ("extent", ctypes.c_float),
                ("lambda_rgb", ctypes.c_float), ("lambda_depth", ctypes.c_float), ("lambda_var", ctypes.c_float),
                ("lambda_flow", ctypes.c_float), ("lambda_still", ctypes.c_float), ("lambda_scale", ctypes.c_float),
                ("lr", ctypes.c_float), ("lr_camera", ctypes.c_float), ("beta1", ctypes.c_float),
                ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("lr_end_factor", ctypes.c_float),
                ("total_iters", _I), ("freeze_rgb", _I), ("freeze_all_splats", _I), ("step_camera", _I)]


def _declare(lib):
    if getattr(lib, "_fit_declared", False):
        return
    lib.gfl_fit_workspace_bytes.restype = ctypes.c_size_t
    lib.gfl_fit_workspace_bytes.argtypes = [ctypes.c_int] * 4
    for name in ("gfl_fit_forward", "gfl_fit_backward_step", "gfl_fit_iteration", "gfl_render_fwd"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(FitState), ctypes.POINTER(FitHyper), _P]
    lib.gfl_fit_iterations.restype = ctypes.c_int
    lib.gfl_fit_iterations.argtypes = [ctypes.POINTER(FitState), ctypes.POINTER(FitHyper), ctypes.c_int, ctypes.c_int, _P]
    lib.gfl_fit_snapshot.restype = ctypes.c_int
    lib.gfl_fit_snapshot.argtypes = [ctypes.POINTER(FitState), ctypes.POINTER(FitHyper), _P, _P, _P, ctypes.c_size_t, _P]
    lib.gfl_fit_iteration_snapshot.restype = ctypes.c_int
    lib.gfl_fit_iteration_snapshot.argtypes = [ctypes.POINTER(FitState), ctypes.POINTER(FitHyper), _P, _P, _P]
    lib.gfl_render_bwd.restype = ctypes.c_int
    lib.gfl_render_bwd.argtypes = [ctypes.POINTER(FitState), ctypes.POINTER(FitHyper), _P, _P, _P, _P, _P, _P]
    lib._fit_declared = True


PROFILE = {"mask": 0}

# Graph captures and graph destruction are serialised process-wide: with several fits in one process
# (fit_video.fit_clips_concurrent) HIP refuses to destroy a graph while ANY stream is capturing ("operation not permitted
# when stream is capturing", raised from a destructor: the process dies).  Launches and replays of other threads are not
# affected (the captures run in "thread_local" mode).
import gc as _gc
import threading as _threading

_GRAPH_LOCK = _threading.RLock()


def set_profile(mask):
    """Enable the library's per-stage HIP-event timing (bit i = stage i, include/gflow_hip.h)."""
    PROFILE["mask"] = int(mask)
    L.check(L.load().gfl_profile_enable(int(mask)), "profile")


class FitEngine:
    def __init__(self, W, H, capacity, device, K_cap=None, bg=0.0, cu_count=0):
        """``cu_count``: compute units the stream this engine is driven on may use (a CU-masked stream, _lib.masked_stream;
        0 = the whole device): the persistent blend grids and their tile queues are sized for it (gfl_fit_state.cu_count)."""
        self.lib = L.load()
        self.cu_count = int(cu_count)
        _declare(self.lib)
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("FitEngine needs a HIP device (there is no CPU path)")
        self.W, self.H = int(W), int(H)
        self.gx, self.gy = (self.W + 15) // 16, (self.H + 15) // 16
        self.T = self.gx * self.gy
        self.N = 0
        self.hp = FitHyper(bg=bg, nearest=0.2, extent=1.3, lambda_rgb=1.0, lambda_depth=0.0, lambda_var=0.0,
                           lambda_flow=0.0, lambda_still=0.0, lambda_scale=0.0, lr=1e-2, lr_camera=0.0, beta1=0.9, beta2=0.999,
                           eps=1e-8, lr_end_factor=0.1, total_iters=0, freeze_rgb=0, freeze_all_splats=0,
                           step_camera=1)
        f32 = dict(dtype=torch.float32, device=self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        H_, W_ = self.H, self.W
        # (fills, not torch.tensor(..., device=): a copy from pageable memory stops the host until the device is idle)
        self.pose = torch.zeros(7, **f32)
        self.pose[3:4].fill_(1.0)
        self.pose_m, self.pose_v = torch.zeros(7, **f32), torch.zeros(7, **f32)
        self.depth_ab = torch.zeros(2, **f32)
        self.depth_ab[0:1].fill_(1.0)
        self.ab_m, self.ab_v = torch.zeros(2, **f32), torch.zeros(2, **f32)
        self.intr = torch.zeros(4, **f32)
        self.extr = torch.zeros(12, **f32)
        self.d_extr = torch.zeros(12, **f32)
        self.step = torch.zeros(1, **i32)
        self.render = torch.zeros(4, H_, W_, **f32)
        self.final_T = torch.zeros(H_, W_, **f32)
        self.n_contrib = torch.zeros(H_, W_, **i32)
        self.d_render = torch.zeros(4, H_, W_, **f32)
        self.err_px = torch.zeros(H_, W_, **f32)
        self.sums = torch.zeros(8, **f32)
        # (one allocation: the pair count tile_offsets[T] and the four overflow words lie side by side -- watch_pending copies
        #  the five of them to the host in ONE copy)
        self._offsets_and_flags = torch.zeros(self.T + 1 + 4, **i32)
        self.tile_offsets = self._offsets_and_flags[:self.T + 1]
        self.tile_range = torch.zeros(self.T, 2, **i32)
        self.overflow = self._offsets_and_flags[self.T + 1:]    # [0]: sticky flag, [1]: iterations that stepped nothing, [2], [3]: this one is void (gflow_hip.h)
        self.gt_rgb = self.gt_depth = self.keep = None
        self.move_mask = self.foot_flags = None
        self.flow_target = self.flow_w = self.still_target = self.still_w = self.row_flags = None
        self.cap = 0
        self.K_cap_req = K_cap
        self._graphs, self._graph_key = {}, None
        self._launched = False
        self._reserved_N = -1          # N of the last full iteration whose last launch reserved the next one's tile regions
        self.busy = False              # checked out by the differentiable operator (gflow_amd.render)
        self._alloc(int(capacity))

    def __del__(self):
        try:
            with _GRAPH_LOCK:
                self._graphs.clear()
                for gs in getattr(self, "_graph_trash", []):
                    gs.clear()
        except Exception:                            # interpreter shutdown
            pass

    def _retire_graphs(self):
        """The captured graphs belong to a state that has changed (new stage, new splat count).  Destroying a graph BLOCKS the
        host until the device is idle (hipGraphExecDestroy: not just until that graph's own replays are done -- destroying
        graphs whose event had long passed still cost the host the whole lead it had over the device, ~5 ms at the first
        iteration of every joint stage: tools/host_lead.py, round 6).  So they are only set aside here; ``reap_graphs`` destroys
        them where the host has just waited for the device anyway (a densification's read, a blocking look at the overflow
        words, the end of a clip)."""
        if self._graphs:
            self.__dict__.setdefault("_graph_trash", []).append(self._graphs)
            self._graphs = {}
        if len(self.__dict__.get("_graph_trash", ())) > 256:         # (a caller that never reaches a reaping point)
            self.reap_graphs()

    def reap_graphs(self):
        """Destroy the graphs set aside by ``_retire_graphs`` -- call where the device is idle (see there).  Under the lock,
        never while a capture runs (_GRAPH_LOCK)."""
        trash = self.__dict__.get("_graph_trash")
        if trash:
            with _GRAPH_LOCK:
                for gs in trash:
                    gs.clear()
            self._graph_trash = []

    # ------------------------------------------------------------------ storage
    def _alloc(self, cap):
        f32 = dict(dtype=torch.float32, device=self.dev)
        old = getattr(self, "params", None)
        n = self.N
        self.cap = cap
        self.params = torch.zeros(cap, ROW, **f32)
        self.adam_m = torch.zeros(cap, ROW, **f32)
        self.adam_v = torch.zeros(cap, ROW, **f32)
        self.rec = torch.zeros(cap, REC, **f32)
        self.d_rec = torch.zeros(cap, REC, **f32) if getattr(self, "want_d_rec", False) else None     # (an optional output)
        if old is not None and n:
            self.params[:n] = old[:n]
        k_min = getattr(self, "K_cap", 0) if getattr(self, "_K_grown", False) else 0      # (never below what an overflow asked for)
        self._alloc_pairs(max(int(self.K_cap_req) if self.K_cap_req else max(4_000_000, 8 * cap), k_min))

    def _alloc_pairs(self, K_cap):
        """The buffers whose size follows K_cap (the sorted ids, and the workspace: keys, per-pair gradient rows, the tile scheduler's state, checkpoints, cached target statistics -- everything in it is rebuilt by the
        next forward / set_targets)."""
        self.K_cap = int(K_cap)                                                         # (64 B per pair: 0.25-0.5 GB of 288)
        self.ids = torch.zeros(self.K_cap, dtype=torch.int32, device=self.dev)
        nbytes = self.lib.gfl_fit_workspace_bytes(self.cap, self.K_cap, self.W, self.H)
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.dev)   # the pool counter must start at 0
        self._state = None
        self._reserved_N = -1

    def ensure_capacity(self, n):
        if n > self.cap:
            regs = (self.flow_target, self.flow_w, self.still_target, self.still_w, self.row_flags, self.foot_flags)
            m, v = self.adam_m, self.adam_v
            old_n = self.N
            self._alloc(max(n, 2 * self.cap))
            self.adam_m[:old_n], self.adam_v[:old_n] = m[:old_n], v[:old_n]
            # per-row side inputs are capacity-sized: re-pad them (the kernels index them by row up to N)
            grown = []
            for t in regs:
                if t is None:
                    grown.append(None)
                    continue
                g = torch.zeros((self.cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.dev)
                g[:t.shape[0]] = t
                grown.append(g)
            self.flow_target, self.flow_w, self.still_target, self.still_w, self.row_flags, self.foot_flags = grown

    def set_count(self, n):
        """Rows [0, n) are live (rows were appended in place behind the old ones)."""
        if n > self.cap:
            raise RuntimeError("FitEngine.set_count beyond the capacity; call ensure_capacity first")
        self.N = int(n)
        if self._state is not None:
            self._state.N = self.N
        self._reserved_N = -1

    def invalidate_regions(self):
        """The host has written pose, intrinsics or parameter rows (through ``views()``, ``pose.copy_`` ...) since the last
        iteration: the tile regions that iteration reserved describe other splats.  Not WRONG to use them -- a tile that
        outgrows its region voids the iteration and it is run again -- but a wasted iteration, predictably at the first
        iteration of every frame and stage; the next iteration bins on the exact path instead."""
        self._reserved_N = -1

    def set_splats(self, attrs):
        """attrs: dict xyz (N,3), scale (N,3), rotate (N,4), opacity (N,1), rgb (N,3) RAW values.
        Copies them into the packed rows (moments are NOT touched)."""
        n = attrs["xyz"].shape[0]
        self.ensure_capacity(n)
        self.N = n
        for k, (a, b) in COLS.items():
            self.params[:n, a:b] = attrs[k].detach().reshape(n, b - a).to(self.dev)
        self.params[:n, 14:] = 0
        self._state = None
        self._reserved_N = -1

    def views(self):
        """Live (N, k) views of the packed parameter rows."""
        return {k: self.params[:self.N, a:b] for k, (a, b) in COLS.items()}

    def reset_optimizer(self, splats=True, camera=True):
        """Fresh Adam state (trainer.py:153 builds a new optimiser on every train call)."""
        if splats:
            self.adam_m.zero_()
            self.adam_v.zero_()
        if camera:
            self.pose_m.zero_(); self.pose_v.zero_(); self.ab_m.zero_(); self.ab_v.zero_()
        self.step.zero_()

    def set_targets(self, gt_image, gt_depth=None, keep=None):
        self.gt_rgb = gt_image.to(self.dev).float().contiguous()
        self.gt_depth = None if gt_depth is None else gt_depth.to(self.dev).float().reshape(self.H, self.W).contiguous()
        self.keep = None if keep is None else keep.to(self.dev).reshape(self.H, self.W).to(torch.uint8).contiguous()
        self.move_mask = self.foot_flags = None
        self._state = None

    def set_footprint_mask(self, move_mask, moving_rows):
        """Camera-only stage (trainer.py:426-451): from now on every forward clears the footprint of
        the splats flagged in ``moving_rows`` from ``keep``, which starts as ``~move_mask`` -- the
        running union the reference builds with ``move_mask = move_gs_mask | move_mask`` in its loop.
        With a non-black background every pixel of the reference's extra render is > 0: all masked."""
        self.move_mask = move_mask.to(self.dev).reshape(self.H, self.W).to(torch.uint8).contiguous()
        flags = torch.zeros(self.cap, dtype=torch.uint8, device=self.dev)
        flags[:moving_rows.shape[0]] = moving_rows.to(self.dev).to(torch.uint8)
        self.foot_flags = flags
        if self.hp.bg > 0.0:
            self.keep = torch.zeros(self.H, self.W, dtype=torch.uint8, device=self.dev)
        else:
            self.keep = (self.move_mask == 0).to(torch.uint8).contiguous()
        self._state = None

    def set_regularisers(self, flow_target=None, flow_w=None, still_target=None, still_w=None, row_flags=None):
        def pad(t, width, dtype):
            if t is None:
                return None
            out = torch.zeros((self.cap, width) if width else (self.cap,), dtype=dtype, device=self.dev)
            out[:t.shape[0]] = t.to(self.dev).reshape((t.shape[0], width) if width else (t.shape[0],))
            return out
        self.flow_target, self.flow_w = pad(flow_target, 2, torch.float32), pad(flow_w, 0, torch.float32)
        self.still_target, self.still_w = pad(still_target, 3, torch.float32), pad(still_w, 0, torch.float32)
        self.row_flags = pad(row_flags, 0, torch.uint8)
        self._state = None

    # ------------------------------------------------------------------- struct
    def state(self):
        if self._state is None:
            p = L.ptr
            s = FitState()
            s.N, s.cap, s.W, s.H, s.K_cap = self.N, self.cap, self.W, self.H, self.K_cap
            for name, t in (("params", self.params), ("adam_m", self.adam_m), ("adam_v", self.adam_v),
                            ("rec", self.rec), ("d_rec", self.d_rec), ("flow_target", self.flow_target),
                            ("flow_w", self.flow_w), ("still_target", self.still_target), ("still_w", self.still_w),
                            ("row_flags", self.row_flags), ("pose", self.pose), ("pose_m", self.pose_m),
                            ("pose_v", self.pose_v), ("depth_ab", self.depth_ab), ("depth_ab_m", self.ab_m),
                            ("depth_ab_v", self.ab_v), ("intr", self.intr), ("extr", self.extr),
                            ("d_extr", self.d_extr), ("step", self.step), ("gt_rgb", self.gt_rgb),
                            ("gt_depth", self.gt_depth), ("keep", self.keep), ("move_mask", self.move_mask),
                            ("foot_flags", self.foot_flags), ("render", self.render),
                            ("final_T", self.final_T), ("n_contrib", self.n_contrib), ("d_render", self.d_render),
                            ("err_px", self.err_px), ("sums", self.sums), ("tile_offsets", self.tile_offsets),
                            ("ids", self.ids), ("tile_range", self.tile_range), ("overflow", self.overflow),
                            ("workspace", self.workspace)):
                setattr(s, name, None if t is None else t.data_ptr())
            s.workspace_bytes = self.workspace.numel()
            s.cu_count = self.cu_count
            if self.gt_rgb is not None and self.foot_flags is None:
                # SSIM statistics of the (masked) target, once per set_targets (gfl_fit_prepare_targets)
                L.check(self.lib.gfl_fit_prepare_targets(ctypes.byref(s), L.stream()), "fit prepare targets")
                s.gt_cached = 1
            self._state = s
        return self._state

    # -------------------------------------------------------------------- calls
    def snapshot_ring(self, rows):
        """(rows, 3, H, W, 3) uint8 on the device: where a train() call keeps its snapshots until its end
        (trainer.make_stepper).  One ring per engine, grown on demand; the current stream waits for the copy that emptied
        it last (``snapshot_ring_copied``)."""
        ring = getattr(self, "_snap_ring", None)
        if ring is None or ring.shape[0] < rows:
            ring = torch.empty(int(rows), 3, self.H, self.W, 3, dtype=torch.uint8, device=self.dev)
            self._snap_ring = ring
        done = getattr(self, "_snap_ring_done", None)
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
        return ring[:rows]

    def snapshot_ring_copied(self):
        """call on the stream that has just been given the ring's device-to-host copy"""
        self._snap_ring_done = torch.cuda.Event()
        self._snap_ring_done.record()

    def forward(self):
        L.check(self.lib.gfl_fit_forward(ctypes.byref(self.state()), ctypes.byref(self.hp), L.stream()), "fit forward")

    def backward_step(self):
        L.check(self.lib.gfl_fit_backward_step(ctypes.byref(self.state()), ctypes.byref(self.hp), L.stream()),
                "fit backward/step")
        self._reserved_N = self.N

    GFL_ITER_RESERVED = 8

    def _reserved_flag(self):
        """GFL_ITER_RESERVED if the coming iteration may bin into the tile regions the last full iteration reserved
        (include/gflow_hip.h): the same splats as then (appended rows land in tiles whose regions were sized without them:
        not wrong -- the library notices a region that is too small and the iteration is run again -- but a wasted
        iteration), and a state the library supports it for."""
        if self._reserved_N != self.N or self.N <= 0:
            return 0
        key = (self.W, self.H, self.K_cap)
        if getattr(self, "_reserved_sup", (None, 0))[0] != key:
            self._reserved_sup = (key, int(self.lib.gfl_fit_reserved_supported(ctypes.byref(self.state()),
                                                                                ctypes.byref(self.hp))))
        return self.GFL_ITER_RESERVED if self._reserved_sup[1] else 0

    def iteration(self, use_graph=False, count=1, snapshot=False, flags=0, reserved=None):
        """``count`` full iterations; ``snapshot=True`` (count 1): the iteration's forward also leaves the three snapshot
        images (gfl_fit_iteration_snapshot: rgb and depth_map_color out of one walk of the lists, center from a small kernel)
        in the engine's own image buffer -- returns that (3, H, W, 3) uint8 tensor, which the NEXT snapshot overwrites.
        ``use_graph=True`` replays a hipGraph of the launches (captured
        lazily, re-captured whenever a pointer, a size or a hyper-parameter changed); it is ignored
        while the library's stage profiler is recording events.  Several iterations in ONE graph save the
        2-6 us that pass between two graph launches (tools/graph_gap.py: 0.2094 -> 0.2028, 0.2079 -> 0.2058 ms per
        iteration with two per graph).  An iteration that follows a full iteration on the same splats bins into the tile
        regions that one reserved (GFL_ITER_RESERVED, include/gflow_hip.h): decided here, a graph per case."""
        if snapshot:
            from .color import lut
            assert count == 1
            if getattr(self, "_snap_out", None) is None:
                self._snap_out = torch.empty(3, self.H, self.W, 3, dtype=torch.uint8, device=self.dev)
            snap_args = (L.ptr(lut("turbo", self.dev)), L.ptr(self._snap_out))
        # (reserved=False: the first iteration of the call takes the exact binning path whatever came before)
        reserved = 0 if (flags or reserved is False or snapshot) else self._reserved_flag()      # (a snapshot iteration bins exactly)
        gkey = ("snap", count, reserved) if snapshot else (count, reserved)
        if use_graph and not PROFILE["mask"] and self._launched and not flags:
            key = bytes(self.state()) + bytes(self.hp)
            if self._graph_key != key:
                self._retire_graphs()                # (the old graphs: destroyed once the device is done with them)
                self._graph_key = key
            g = self._graphs.get(gkey)
            if g is None:
                # capture_begin / capture_end directly: the torch.cuda.graph() context manager synchronises the device,
                # runs the Python garbage collector and empties the allocator's cache on entry -- a full stop of the fit,
                # ~60 times per 8-frame clip (the graphs are re-captured whenever N or a hyper-parameter changes) -- to
                # protect allocations inside the capture; the library allocates nothing.  "thread_local": a capture
                # in one host thread must not fail because another thread (another clip on the same device,
                # fit_video.fit_clips_concurrent) reads a value back at that moment.
                if getattr(self, "_capture_stream", None) is None:
                    self._capture_stream = torch.cuda.Stream(device=self.dev)
                side = self._capture_stream
                cur = torch.cuda.current_stream()
                st, hp = self.state(), self.hp
                with _GRAPH_LOCK:
                    gc_on = _gc.isenabled()
                    _gc.disable()                    # (a cycle collection inside the capture could destroy an old engine's graphs)
                    try:
                        g = torch.cuda.CUDAGraph()
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            g.capture_begin(capture_error_mode="thread_local")
                            try:
                                if snapshot:
                                    L.check(self.lib.gfl_fit_iteration_snapshot(ctypes.byref(st), ctypes.byref(hp), *snap_args,
                                                                                L.stream()), "snapshot iteration (capture)")
                                else:
                                    L.check(self.lib.gfl_fit_iterations(ctypes.byref(st), ctypes.byref(hp), count, reserved,
                                                                        L.stream()), "fit iterations (capture)")
                            finally:
                                g.capture_end()
                        cur.wait_stream(side)
                    finally:
                        if gc_on:
                            _gc.enable()
                self._graphs[gkey] = g
            g.replay()
            self._reserved_N = self.N
            return self._snap_out if snapshot else None
        if snapshot:
            L.check(self.lib.gfl_fit_iteration_snapshot(ctypes.byref(self.state()), ctypes.byref(self.hp), *snap_args, L.stream()),
                    "snapshot iteration")
        else:
            L.check(self.lib.gfl_fit_iterations(ctypes.byref(self.state()), ctypes.byref(self.hp), count, int(flags) | reserved,
                                                L.stream()), "fit iterations")
        self._reserved_N = self.N
        self._launched = True          # every kernel is loaded now: capture is safe from here on
        return self._snap_out if snapshot else None

    def snapshot(self, out=None):
        """(3, H, W, 3) uint8 on the device: rgb, depth_map_color, center of the last forward (gfl_fit_snapshot).
        Only AFTER the iteration's backward.  ``out``: a contiguous (3, H, W, 3) uint8 tensor to write into."""
        from .color import lut
        need = self.lib.gfl_fit_snapshot_workspace_bytes(self.N, self.W, self.H)
        if getattr(self, "_snap_ws", None) is None or self._snap_ws.numel() < need:
            self._snap_ws = torch.empty(int(need * 1.5), dtype=torch.uint8, device=self.dev)
        if out is None:
            out = torch.empty(3, self.H, self.W, 3, dtype=torch.uint8, device=self.dev)
        L.check(self.lib.gfl_fit_snapshot(ctypes.byref(self.state()), ctypes.byref(self.hp), L.ptr(lut("turbo", self.dev)),
                                          L.ptr(out), L.ptr(self._snap_ws), self._snap_ws.numel(), L.stream()), "snapshot")
        return out

    # ------------------------------------------------------------- saved states
    _STATE_TENSORS = ("step", "pose", "pose_m", "pose_v", "depth_ab", "ab_m", "ab_v")

    def save_state(self):
        """A copy of everything an iteration changes and the next one reads: the live rows with their Adam moments, the
        camera and depth-affine parameters with theirs, the step counter.  (bench.py times the SAME window of a fit
        over and over: ``restore_state`` puts the fit back at the window's first iteration.)"""
        n = self.N
        out = {"N": n, "rows": [t[:n].clone() for t in (self.params, self.adam_m, self.adam_v)]}
        for k in self._STATE_TENSORS:
            out[k] = getattr(self, k).clone()
        return out

    def restore_state(self, saved):
        """Back to ``save_state``'s moment, on the current stream.  Elementwise kernels, not ``copy_``: device-to-device
        ``copy_`` goes through the runtime's blit kernel (47 us for 3.7 MB, trainer.py on the snapshot ring)."""
        n = saved["N"]
        if n != self.N:
            raise RuntimeError("FitEngine.restore_state: the splat count has changed since save_state")
        self._reserved_N = -1          # (the regions were sized for the splats as they are now, not as they were then)
        for dst, src in zip((self.params, self.adam_m, self.adam_v), saved["rows"]):
            torch.bitwise_or(src.view(torch.int32), 0, out=dst[:n].view(torch.int32))
        for k in self._STATE_TENSORS:
            t = getattr(self, k)
            torch.bitwise_or(saved[k].view(torch.int32), 0, out=t.view(torch.int32))

    # ------------------------------------------------------------------ outputs
    @property
    def uv(self):
        return self.rec[:self.N, 0:2]

    @property
    def depth(self):
        return self.rec[:self.N, 9:10]

    @property
    def K(self):
        return int(self.tile_offsets[self.T].item())

    def schedule(self, forward=False):
        """The tile queues of the last forward as a list of 1-D int64 tensors (tile ids per queue): the backward
        blend's, or (``forward=True``) the forward blend's."""
        nq, cap = ctypes.c_int(), ctypes.c_int()
        lists, counts = ctypes.c_void_p(), ctypes.c_void_p()
        fn = self.lib.gfl_fit_schedule_info_fwd if forward else self.lib.gfl_fit_schedule_info
        L.check(fn(ctypes.byref(self.state()), ctypes.byref(nq), ctypes.byref(cap),
                   ctypes.byref(lists), ctypes.byref(counts)), "fit schedule info")
        off_l = lists.value - self.workspace.data_ptr()
        off_c = counts.value - self.workspace.data_ptr()
        torch.cuda.synchronize(self.dev)
        lst = self.workspace[off_l:off_l + 4 * nq.value * cap.value].view(torch.int32).reshape(nq.value, cap.value).cpu()
        cnt = self.workspace[off_c:off_c + 4 * nq.value].view(torch.int32).cpu()
        return [(lst[c, :int(cnt[c])] & 0xffff).long() for c in range(nq.value)]

    def check_overflow(self):
        """Blocking read of the pair-list overflow flag (sticky on the device); raises if pairs were dropped."""
        self._ovf_event = None
        code = int(self.overflow[0].item())
        self.reap_graphs()                           # (the device is idle here)
        if code:
            self.overflow.zero_()
            raise RuntimeError(self._overflow_message(code))

    def settle_overflow(self):
        """Blocking.  If a forward dropped (splat, tile) pairs since the last call: the lists are doubled, both words are
        cleared, and the number of iterations that stepped NOTHING meanwhile is returned (the library skips every update
        while the flag is set, include/gflow_hip.h) -- the caller runs that many iterations again and the fit is where a
        fit that never overflowed would be.  The same count, without anything to grow, for iterations that were void
        because a tile outgrew its reserved region.  0: nothing happened.  (K_cap is max(4 M, 8 x capacity), ~15 x what fits
        produce: this is the rare path, but a silent or fatal one it must not be.)"""
        self._ovf_event = None
        self._pend_event = None        # (a blocking look supersedes an outstanding watch: its words are read and cleared here)
        code, skipped = (int(v) for v in self.overflow[:2].tolist())                   # the host read
        self.reap_graphs()                           # (the device is idle here)
        if code == 0:
            if skipped > 0:
                # tiles outgrew the regions reserved for them in `skipped` iterations (GFL_ITER_RESERVED): those stepped
                # nothing, the ones after them were fine again -- nothing to grow, that many iterations more to run
                self.overflow[1:2].zero_()
                self.regions_outgrown = getattr(self, "regions_outgrown", 0) + skipped
            return max(skipped, 0)
        if code != 1:
            self.overflow.zero_()
            raise RuntimeError(self._overflow_message(code))
        self._K_grown = True
        self._alloc_pairs(2 * self.K_cap)
        self.overflow.zero_()
        with _GRAPH_LOCK:
            self._graphs.clear()
        self._graph_key = None
        self.pairs_grown = getattr(self, "pairs_grown", 0) + 1
        return max(skipped, 0)

    def grow_pairs(self):
        """Twice the room for (splat, tile) pairs, before they run out (the caller has drained the stream)."""
        self._pend_event = None
        self._K_grown = True
        self._alloc_pairs(2 * self.K_cap)
        with _GRAPH_LOCK:
            self._graphs.clear()
        self._graph_key = None
        self.pairs_grown = getattr(self, "pairs_grown", 0) + 1

    def _overflow_message(self, code):
        if code == 2:
            return ("FitEngine: an iteration was told that the previous one had run its preprocess / reserved its tile regions, "
                    "and it had not")
        return f"FitEngine: more than K_cap={self.K_cap} splat-tile pairs; raise K_cap"

    def watch_pending(self):
        """Non-blocking: the four overflow words and the pair count as they are once everything queued so far has run, copied
        to page-locked memory behind that work.  ``read_pending`` waits for exactly that copy -- not for work queued after
        this call: the trainer queues one more iteration first, so the device is busy while the host looks.
        The pair count is ``tile_offsets[T]``: the exact binning path writes it (gfl_fit_bin.hip, the scatter's scan) and so
        does an iteration on reserved regions -- the region-reserving workgroup of its LAST launch stores the sum of the
        tile counts there (gfl_fit_order.hpp: ``*ro.total = pairs``, ``ro.total = tile_offsets + T`` in gfl_fit.hip) -- so the
        word is at most one iteration old, whichever path ran (ADVICE r05 asked whether the reserved path leaves it stale)."""
        if getattr(self, "_pend_host", None) is None:
            self._pend_host = torch.zeros(8, dtype=torch.int32, pin_memory=True)
        self._pend_host[0:5].copy_(self._offsets_and_flags[self.T:self.T + 5], non_blocking=True)      # K, then the four words
        self._pend_event = torch.cuda.Event()
        self._pend_event.record()

    def pending_ready(self):
        """False while a ``watch_pending`` copy has not landed yet (``read_pending`` would wait)."""
        ev = getattr(self, "_pend_event", None)
        return ev is None or ev.query()

    def read_pending(self):
        """(code, iterations that stepped nothing, pair count) of the last ``watch_pending``; None without one."""
        ev = getattr(self, "_pend_event", None)
        if ev is None:
            return None
        ev.synchronize()
        self._pend_event = None
        v = self._pend_host.tolist()
        return v[1], v[2], v[0]

    def watch_overflow(self):
        """The same check without stopping the host: the flag is copied to pinned memory behind the work queued so
        far; ``poll_overflow`` (called here first, for the previous watch) raises once that copy has landed and shows
        dropped pairs.  train() calls this once per frame -- a blocking read there drained the queue between two
        stages -- and fit_clip ends with the blocking check."""
        self.poll_overflow()
        if getattr(self, "_ovf_host", None) is None:
            self._ovf_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        self._ovf_host.copy_(self.overflow[0:1], non_blocking=True)
        self._ovf_event = torch.cuda.Event()
        self._ovf_event.record()

    def poll_overflow(self):
        ev = getattr(self, "_ovf_event", None)
        if ev is not None and ev.query():
            self._ovf_event = None
            if int(self._ovf_host[0]):
                self.overflow.zero_()
                raise RuntimeError(self._overflow_message(int(self._ovf_host[0])))

    def loss_terms(self):
        """(loss_rgb, loss_depth) of the last backward_step as device scalars (trainer.py:460-462,485)."""
        hw = float(self.H * self.W)
        return self.sums[0] / hw + (1.0 - self.sums[1] / (3.0 * hw)), self.sums[2] / hw
