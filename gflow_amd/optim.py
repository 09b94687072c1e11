"""Fused Adam + LinearLR for the fit loop (gflow/trainer.py:123-153,383-384,554-555).

Keeps the torch.optim surface the trainer uses (param_groups with per-group lr,
``zero_grad``/``step``, a scheduler object with ``step``) but every group is ONE HIP
kernel launch, the step counter and the LinearLR factor live on the device (so an
iteration can be captured in a hipGraph and replayed), and rows of a group can be
frozen with a per-row mask (trainer.py:543-546 zeroes xyz grads of still splats).
"""
import torch

from . import _lib as L


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if isinstance(params, (list, tuple)) and len(params) and isinstance(params[0], dict):
            groups = [dict(g) for g in params]
        else:
            groups = [{"params": list(params)}]
        self.param_groups = []
        for g in groups:
            ps = g["params"]
            ps = [ps] if isinstance(ps, torch.Tensor) else list(ps)
            g = dict(g, params=ps)
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            self.param_groups.append(g)
        self.state = {}
        dev = None
        for g in self.param_groups:
            for p in g["params"]:
                L.need_device(p)
                dev = p.device
        self._dev = dev
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev) if dev is not None else None
        self.lr_end_factor = 1.0
        self.lr_total_iters = 0
        self.row_zero = {}           # id(param) -> uint8 mask, rows whose grad is forced to 0

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def set_row_zero_grad(self, param, mask):
        """mask (rows,) bool/uint8: 1 = this row of ``param`` receives a zero gradient."""
        self.row_zero[id(param)] = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()

    @torch.no_grad()
    def step(self):
        lib = L.load()
        stepped = False
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(id(p))
                if st is None:
                    st = self.state[id(p)] = {"exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                if not p.is_contiguous():
                    raise RuntimeError("gflow_amd.optim.Adam: parameters must be contiguous")
                grad = p.grad.contiguous()
                rz = self.row_zero.get(id(p))
                row_len = p.shape[-1] if p.dim() > 1 else 1
                L.check(lib.gfl_adam_step(L.ptr(p), L.ptr(grad), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]),
                                          p.numel(), row_len, L.ptr(rz), float(g["lr"]), b1, b2, g["eps"],
                                          L.ptr(self.step_count), float(self.lr_end_factor),
                                          int(self.lr_total_iters), L.stream()), "adam")
                stepped = True
        if stepped:
            L.check(lib.gfl_step_increment(L.ptr(self.step_count), L.stream()), "adam step counter")


class LinearLR:
    """torch.optim.lr_scheduler.LinearLR(start_factor=1.0, end_factor, total_iters):
    the factor is evaluated on the device from the optimiser's own step counter, so
    ``step()`` is a no-op kept for loop-shape parity (trainer.py:555)."""

    def __init__(self, optimizer, start_factor=1.0, end_factor=0.1, total_iters=1):
        if start_factor != 1.0:
            raise ValueError("only start_factor=1.0 is supported (that is what trainer.py:384 uses)")
        self.optimizer = optimizer
        optimizer.lr_end_factor = float(end_factor)
        optimizer.lr_total_iters = int(total_iters)

    def step(self):
        pass
