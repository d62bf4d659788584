"""The optimisation step of ``Trainer._backward_step`` (reference training/trainer.py:219-247) as one HIP kernel.

The reference runs ``nan_to_num`` per gradient, ``torch.optim.AdamW.step()`` (about ten ``_foreach`` passes) and a
``lerp`` per EMA tensor: three sweeps over 226 M parameters and their moments.  ``FusedAdamEMA`` drives
``swiftk_adamw_ema_step`` instead: one pass that sanitises the gradient, applies the Adam / AdamW rule with the
optimizer's own hyper-parameters (``param_groups`` stay the source of truth: the LR schedule writes ``g["lr"]`` there)
and updates the EMA copy.  The torch optimizer object keeps owning the state: ``exp_avg`` / ``exp_avg_sq`` of every
parameter are views of two flat buffers and ``step`` is a shared counter tensor, so ``optimizer.state_dict()`` /
``load_state_dict()`` and the checkpoint format (trainer.py:522-535) are unchanged.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from .._lib import OPT_MAX_GROUPS, OptChunk, OptHyper, SwiftkError, check, lib

CHUNK = 16384


def supported(optimizer) -> bool:
    """Plain Adam / AdamW on device parameters (no amsgrad / maximize / capturable / differentiable variants)."""
    if type(optimizer) not in (torch.optim.Adam, torch.optim.AdamW):
        return False
    gs = optimizer.param_groups
    if not 1 <= len(gs) <= OPT_MAX_GROUPS:
        return False
    b, e = gs[0]["betas"], gs[0]["eps"]
    for g in gs:
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return False
        if tuple(g["betas"]) != tuple(b) or g["eps"] != e:  # one (beta1, beta2, eps) per launch
            return False
        if any((not p.is_cuda) or p.dtype != torch.float32 for p in g["params"]):
            return False
    return True


class FusedAdamEMA:
    def __init__(self, optimizer, net_params: Sequence[torch.nn.Parameter], ema_params: Optional[Sequence[torch.Tensor]],
                 grad_flat: torch.Tensor):
        """``net_params`` in the order of ``grad_flat`` (every ``param.grad`` is a view of it); ``ema_params`` aligned with
        ``net_params`` (or None)."""
        if not supported(optimizer):
            raise SwiftkError("FusedAdamEMA needs torch.optim.Adam / AdamW on fp32 device parameters")
        self.opt = optimizer
        self.decoupled = type(optimizer) is torch.optim.AdamW
        dev = grad_flat.device
        group_of = {}
        for gi, g in enumerate(optimizer.param_groups):
            for p in g["params"]:
                group_of[id(p)] = gi
        n = grad_flat.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_t = torch.zeros((), dtype=torch.float32)
        chunks, off = [], 0
        self._keep = []
        for i, p in enumerate(net_params):
            if id(p) not in group_of:
                raise SwiftkError("every trainable parameter must be in one of the optimizer's param_groups")
            if not p.is_contiguous():
                raise SwiftkError("parameters must be contiguous")
            e = None if ema_params is None else ema_params[i]
            if e is not None and (e.shape != p.shape or not e.is_contiguous() or e.dtype != torch.float32):
                raise SwiftkError("EMA tensors must mirror the parameters (shape, fp32, contiguous)")
            st = optimizer.state.get(p, {})
            mv, vv = self.m[off:off + p.numel()].view_as(p), self.v[off:off + p.numel()].view_as(p)
            if "exp_avg" in st:  # resumed run: adopt the loaded moments (trainer.py:104-116)
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
                self.step_t.fill_(float(st["step"]))
            optimizer.state[p] = {"step": self.step_t, "exp_avg": mv, "exp_avg_sq": vv}
            for s in range(0, p.numel(), CHUNK):
                k = min(CHUNK, p.numel() - s)
                chunks.append((p.data_ptr() + 4 * s, 0 if e is None else e.data_ptr() + 4 * s, off + s, k, group_of[id(p)]))
            off += p.numel()
        assert off == n
        arr = (OptChunk * len(chunks))()
        for j, (pp, ee, fo, k, gi) in enumerate(chunks):
            arr[j].p, arr[j].ema, arr[j].flat_off, arr[j].n, arr[j].group = pp, ee or None, fo, k, gi
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = host.to(dev)
        self.n_chunks = len(chunks)
        self.grad_flat = grad_flat
        self._touched = tuple(net_params) + tuple(e for e in (ema_params or ()) if e is not None)

    def step(self, ema_beta: float) -> None:
        """One fused step with the CURRENT ``param_groups`` hyper-parameters; bumps the parameters' version counters so
        the engines rebuild their GEMM-operand copies."""
        self.step_t += 1
        t = float(self.step_t)
        g0 = self.opt.param_groups[0]
        b1, b2 = g0["betas"]
        h = OptHyper()
        for gi, g in enumerate(self.opt.param_groups):
            h.lr[gi], h.weight_decay[gi] = float(g["lr"]), float(g["weight_decay"])
            h.step_size[gi] = float(g["lr"]) / (1.0 - b1 ** t)
        h.beta1, h.beta2, h.eps = float(b1), float(b2), float(g0["eps"])
        h.bias2_sqrt = (1.0 - b2 ** t) ** 0.5
        h.ema_beta = float(ema_beta)
        h.decoupled = int(self.decoupled)
        check(lib().swiftk_adamw_ema_step(self.table.data_ptr(), self.n_chunks, self.grad_flat.data_ptr(), self.m.data_ptr(),
                                          self.v.data_ptr(), C.byref(h), torch.cuda.current_stream().cuda_stream),
              "swiftk_adamw_ema_step")
        # the kernel wrote through raw pointers: bump the version counters torch would have bumped, so the engines
        # (SwinEngine.refresh and friends compare (data_ptr, _version) stamps) rebuild their GEMM-operand copies
        ts = self._touched
        torch._C._autograd._unsafe_set_version_counter(ts, tuple(t._version + 1 for t in ts))
