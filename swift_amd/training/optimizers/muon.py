"""MuonWithAuxAdam on the gfx950 GEMM (mirrors reference src/swift/training/optimizers/muon.py:157-264).

Same constructor contract (param groups flagged ``use_muon``; same defaults and key check), same update rule:
Nesterov momentum -> quintic Newton-Schulz orthogonalisation in bf16 (5 iterations) -> ``sqrt(max(1, rows/cols))`` rescale ->
decoupled weight decay -> step; AdamW-style update for the non-Muon groups; round-robin parameter ownership across ranks as
in the reference, the updated parameters travelling by broadcast from their owner (RCCL over xGMI; shape-agnostic, where
the reference's per-round all-gather needs equal shapes within a round).

The three products of every Newton-Schulz iteration (``X X^T``, ``A A``, ``B X`` -- all the FLOPs of the optimiser: 5 x 3
GEMMs on up to 5632 x 1056 matrices per parameter) run on ``swiftk_gemm`` (bf16 MFMA, bf16 results like the reference's bf16
matmuls); the O(n) element-wise glue (momentum lerp, ``bA + cA^2``, ``aX + BX``, Frobenius norm) is torch on device tensors.
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist

from ... import ops
from ..._lib import EPI_NONE, check, lib

_BF = torch.bfloat16
NS_COEFFS = (3.4445, -4.7750, 2.0315)


def _s():
    return torch.cuda.current_stream().cuda_stream


_SLABS = {}


def _gemm_bf16(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """out[:, :n] = a[:, :k] @ w[:n, :k]^T (bf16 operands with zero K padding, bf16 result).

    The products of the orthogonaliser have few output tiles (1056 x 1056 = 15 tiles of 256 x 352 on 256 CUs) and long
    contractions, so they go through the split-K form of the GEMM (k-ranges into fp32 slabs, summed by
    ``swiftk_reduce_slabs``) whenever that fills more of the chip; the sum is rounded to bf16 once, like a bf16 matmul."""
    m = a.shape[0]
    tiles = ((m + 255) // 256) * ((n + 351) // 352)
    ks = max(1, min(32, 256 // tiles, k // 128))
    if ks > 1 and n % 8 == 0 and m % 8 == 0:
        need = ks * m * n
        dev = a.device
        if dev not in _SLABS or _SLABS[dev].numel() < need + m * n:
            _SLABS[dev] = torch.empty(need + m * n, dtype=torch.float32, device=dev)
        slabs, acc = _SLABS[dev][:need], _SLABS[dev][need:need + m * n].view(m, n)
        check(lib().swiftk_gemm_splitk(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), slabs.data_ptr(), n, m * n, m, n, k,
                                       ops.dtype_code(_BF), ks, _s()), "swiftk_gemm_splitk")
        check(lib().swiftk_reduce_slabs(slabs.data_ptr(), n, m * n, ks, acc.data_ptr(), n, m, n, 0, _s()), "swiftk_reduce_slabs")
        out[:, :n] = acc
        return out
    check(lib().swiftk_gemm(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), a.shape[0], n,
                            k, ops.dtype_code(_BF), ops.dtype_code(_BF), EPI_NONE, None, None, 0, _s()), "swiftk_gemm")
    return out


def _transpose_into(src: torch.Tensor, rows: int, cols: int, dst: torch.Tensor) -> torch.Tensor:
    check(lib().swiftk_transpose(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), rows, cols, ops.dtype_code(_BF),
                                 _s()), "swiftk_transpose")
    return dst


def zeropower_via_newtonschulz5(G: torch.Tensor, steps: int = 5) -> torch.Tensor:
    """Quintic Newton-Schulz iteration X <- aX + (bA + cA^2)X, A = XX^T, on G / |G|_F (muon.py:5-35) for one 2-D matrix on the
    GPU; wide orientation inside (rows <= cols), bf16 throughout, returns bf16 of G's shape."""
    assert G.ndim == 2 and G.is_cuda
    a, b, c = NS_COEFFS
    tall = G.shape[0] > G.shape[1]
    X = G.to(_BF).mT if tall else G.to(_BF)
    X = X / (X.norm() + 1e-7)  # Frobenius norm, in bf16 as the reference takes it
    m, n = X.shape
    if m % 4 or n % 4 or m < 16:
        # degenerate "matrices" the reference's rule also sends through Muon (the [1, heads, 1, 1] logit scale viewed as
        # 1 x heads): a few dozen FLOPs, below the GEMM's shape granularity -> plain library matmul
        for _ in range(steps):
            gram = X @ X.mT
            X = a * X + (b * gram + c * (gram @ gram)) @ X
        return X.mT if tall else X
    km, kn = ops.k_pad(_BF, m), ops.k_pad(_BF, n)
    dev = G.device
    Xb = torch.zeros(m, kn, dtype=_BF, device=dev)
    Xb[:, :n] = X
    XT = torch.zeros(n, km, dtype=_BF, device=dev)
    A, A2 = torch.zeros(m, km, dtype=_BF, device=dev), torch.zeros(m, km, dtype=_BF, device=dev)
    BX = torch.zeros(m, kn, dtype=_BF, device=dev)
    for _ in range(steps):
        _gemm_bf16(Xb, Xb, A, m, kn)                 # A = X X^T
        _gemm_bf16(A, A, A2, m, km)                  # A A   (A symmetric)
        B = b * A + c * A2
        _transpose_into(Xb, m, n, XT)                # operand form of X for B X
        _gemm_bf16(B, XT, BX, n, km)                 # B X
        Xb = a * Xb + BX
    X = Xb[:, :n]
    return X.mT if tall else X


def _gemm_stack(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """out[l][:, :n] = a[l][:, :k] @ w[l][:n, :k]^T for every matrix l of three [L, rows, ld] bf16 stacks: ONE launch
    (``swiftk_gemm_batched``).  Twelve same-shape products fill the chip by themselves (180 output tiles for the 1056-wide
    squares), where a single one needs the split-K slabs and the reduction pass of ``_gemm_bf16``."""
    check(lib().swiftk_gemm_batched(a.data_ptr(), a.stride(1), a.stride(0), w.data_ptr(), w.stride(1), w.stride(0), out.data_ptr(),
                                    out.stride(1), out.stride(0), a.shape[0], a.shape[1], n, k, ops.dtype_code(_BF),
                                    ops.dtype_code(_BF), _s()), "swiftk_gemm_batched")
    return out


def _fro_norm(X: torch.Tensor) -> torch.Tensor:
    """Frobenius norm per matrix of a bf16 stack [L, m, n] -> [L, 1, 1] bf16, summed in fp32 and rounded once like ATen's bf16
    norm -- but in two stages over 1,024 slices per matrix: ATen reduces each matrix with ONE workgroup (870 us for the twelve
    1056 x 5632 matrices of w1, 30 us this way; tools/norm_probe.py)."""
    L, m, n = X.shape
    if (m * n) % 1024 or not X.is_contiguous():
        return X.norm(dim=(-2, -1), keepdim=True)
    part = torch.linalg.vector_norm(X.view(L, 1024, -1), dim=2, dtype=torch.float32)
    return torch.linalg.vector_norm(part, dim=1).to(X.dtype).view(L, 1, 1)


def zeropower_stack(G: torch.Tensor, steps: int = 5) -> torch.Tensor:
    """``zeropower_via_newtonschulz5`` for a stack [L, rows, cols] of same-shape matrices at once (the reference's routine
    takes batches as well, muon.py:13-35): per iteration three batched GEMM launches, L transposes and four element-wise
    launches for the whole stack instead of that many per matrix.  rows, cols multiples of 4, min(rows, cols) >= 16."""
    assert G.ndim == 3 and G.is_cuda
    a, b, c = NS_COEFFS
    tall = G.shape[1] > G.shape[2]
    X = G.to(_BF)
    X = X / (_fro_norm(X) + 1e-7)  # (the norm of a matrix is that of its transpose: taken on the contiguous form)
    if tall:
        X = X.mT
    L, m, n = X.shape
    assert m % 4 == 0 and n % 4 == 0 and m >= 16
    km, kn = ops.k_pad(_BF, m), ops.k_pad(_BF, n)
    dev = G.device
    # the stack with every matrix's rows padded to km (zero rows): transposing it as ONE [L km, n] matrix then yields, side by
    # side in [n, L km], the k-padded operand forms X_l^T of all matrices -- one launch per iteration instead of L
    Xp = torch.zeros(L, km, kn, dtype=_BF, device=dev)
    Xb = Xp[:, :m, :]
    Xb[:, :, :n] = X
    XTall = torch.empty(n, L * km, dtype=_BF, device=dev)
    XT = XTall.view(n, L, km).permute(1, 0, 2)       # [L, n, km]: matrix l at column offset l km, row stride L km
    A, A2 = torch.zeros(L, m, km, dtype=_BF, device=dev), torch.zeros(L, m, km, dtype=_BF, device=dev)
    BX = torch.zeros(L, m, kn, dtype=_BF, device=dev)
    for _ in range(steps):
        _gemm_stack(Xb, Xb, A, m, kn)                # A = X X^T
        _gemm_stack(A, A, A2, m, km)                 # A A   (A symmetric)
        B = A2.mul_(c).add_(A.mul_(b))               # b A + c A^2 with the reference's three bf16 roundings, in place (A is rebuilt next trip)
        _transpose_into(Xp.view(L * km, kn), L * km, n, XTall)  # operand forms of X for B X
        _gemm_stack(B, XT, BX, n, km)                # B X
        Xb.mul_(a).add_(BX)
    X = Xb[:, :, :n]
    return X.mT if tall else X


def zeropower_stack_small(G: torch.Tensor, steps: int = 5) -> torch.Tensor:
    """The same iteration for a stack [L, rows, cols] of matrices below the GEMM's shape granularity (the twelve layers'
    [1, heads, 1, 1] logit scales viewed as 1 x heads, which the reference's rule also sends through Muon): batched library
    matmuls on the whole stack -- a few dozen launches instead of ~100 per matrix."""
    a, b, c = NS_COEFFS
    tall = G.shape[1] > G.shape[2]
    X = G.to(_BF).mT if tall else G.to(_BF)
    X = X / (X.norm(dim=(-2, -1), keepdim=True) + 1e-7)
    for _ in range(steps):
        gram = X @ X.mT
        X = a * X + (b * gram + c * (gram @ gram)) @ X
    return X.mT if tall else X


def muon_update(grad: torch.Tensor, momentum: torch.Tensor, beta: float = 0.95, ns_steps: int = 5, nesterov: bool = True):
    """The Muon direction of one parameter (muon.py:38-45): gradient EMA, Nesterov look-ahead (written into ``grad``, which the
    reference consumes the same way), orthogonalisation of the matrix view, sqrt(max(1, rows / cols)) rescale."""
    momentum.mul_(beta).add_(grad, alpha=1.0 - beta)
    direction = grad.mul_(1.0 - beta).add_(momentum, alpha=beta) if nesterov else momentum
    matrix = direction.flatten(1) if direction.ndim == 4 else direction
    ortho = zeropower_via_newtonschulz5(matrix, steps=ns_steps)  # (module-level lookup: tests swap in a CPU orthogonaliser)
    return ortho * math.sqrt(max(1.0, grad.shape[-2] / grad.shape[-1]))


def adam_update(grad, buf1, buf2, step, betas, eps):
    """Bias-corrected Adam direction (muon.py:149-154); ``buf1`` / ``buf2`` are the running first / second moments."""
    b1, b2 = betas
    buf1.mul_(b1).add_(grad, alpha=1.0 - b1)
    buf2.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
    return (buf1 / (1.0 - b1 ** step)) / ((buf2 / (1.0 - b2 ** step)).sqrt_() + eps)


# per-group hyper-parameters the reference fills in (muon.py:172-184); the key set of a group is checked against them
_GROUP_DEFAULTS = {
    True: {"lr": 0.02, "momentum": 0.95, "weight_decay": 0},
    False: {"lr": 3e-4, "betas": (0.9, 0.95), "eps": 1e-10, "weight_decay": 0},
}


class MuonWithAuxAdam(torch.optim.Optimizer):
    """Param groups flagged ``use_muon`` (2-D+ transformer weights) take the Muon step, the others an AdamW-style step.  State
    keys as in the reference (``momentum_buffer`` / ``exp_avg``, ``exp_avg_sq``, ``step``) so optimiser checkpoints carry over."""

    def __init__(self, param_groups, **kwargs):
        groups = []
        for given in param_groups:
            if "use_muon" not in given:
                raise AssertionError("every param group must say use_muon=True/False (muon.py:170)")
            kind = bool(given["use_muon"])
            group = {**_GROUP_DEFAULTS[kind], **given}
            if set(group) != set(_GROUP_DEFAULTS[kind]) | {"params", "use_muon"}:
                raise AssertionError(f"unexpected keys in a {'Muon' if kind else 'Adam'} group: {sorted(group)}")
            if kind:  # largest first: the rounds of the rank round-robin then hold parameters of similar cost
                group["params"] = sorted(group["params"], key=lambda q: q.size(), reverse=True)
            groups.append(group)
        super().__init__(groups, dict())

    def _muon_group(self, group, world: int, rank: int, multi: bool) -> None:
        params = group["params"]
        decay = 1.0 - group["lr"] * group["weight_decay"]
        for base in range(0, len(params), world):  # round `base // world`: rank r owns params[base + r]
            mine = base + rank
            if mine < len(params):
                p = params[mine]
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                st = self.state[p]
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p)
                step = muon_update(p.grad, st["momentum_buffer"], beta=group["momentum"])
                p.mul_(decay).add_(step.reshape(p.shape).to(p.dtype), alpha=-group["lr"])
            if multi:
                # the reference all-gathers the round's parameters (muon.py:234-237), which needs one shape per round;
                # rounds here may mix shapes (12 x w1 then 12 x to_qkv on 8 ranks), so every parameter of the round is
                # broadcast from the rank that owns it
                for r in range(min(world, len(params) - base)):
                    dist.broadcast(params[base + r].data, src=r)

    def _muon_group_stacked(self, group) -> None:
        """One rank, parameters on the GPU: the same update with the same-shape matrices of the group (the twelve layers' w1,
        w2, to_qkv, wo) walked TOGETHER -- multi-tensor launches for the momentum / Nesterov blend and the parameter update, the
        stacked orthogonaliser in between.  Issued matrix by matrix the step is ~5,000 launches of 5-25 us (65 ms at Swift-B)."""
        beta, lr = group["momentum"], group["lr"]
        decay = 1.0 - lr * group["weight_decay"]
        classes = {}
        for p in group["params"]:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            if "momentum_buffer" not in self.state[p]:
                self.state[p]["momentum_buffer"] = torch.zeros_like(p)
            classes.setdefault(tuple(p.shape), []).append(p)
        for shape, ps in classes.items():
            rows, cols = shape[0], max(1, math.prod(shape[1:])) if len(shape) == 4 else (shape[-1] if len(shape) > 1 else 1)
            small = rows % 4 != 0 or cols % 4 != 0 or min(rows, cols) < 16  # below the GEMM's granularity: library matmuls
            if len(ps) < 2 or len(shape) not in (2, 4):
                for p in ps:  # lone shapes: matrix by matrix
                    step = muon_update(p.grad, self.state[p]["momentum_buffer"], beta=beta)
                    p.mul_(decay).add_(step.reshape(p.shape).to(p.dtype), alpha=-lr)
                continue
            grads, bufs = [p.grad for p in ps], [self.state[p]["momentum_buffer"] for p in ps]
            torch._foreach_mul_(bufs, beta)                       # gradient EMA ...
            torch._foreach_add_(bufs, grads, alpha=1.0 - beta)
            torch._foreach_mul_(grads, 1.0 - beta)                # ... and the Nesterov look-ahead, written into the gradients
            torch._foreach_add_(grads, bufs, alpha=beta)
            stack = torch.stack([g.reshape(rows, cols) for g in grads])
            ortho = zeropower_stack_small(stack) if small else zeropower_stack(stack)
            scaled = (ortho * math.sqrt(max(1.0, shape[-2] / shape[-1]))).float()
            torch._foreach_mul_(ps, decay)
            torch._foreach_add_(ps, [u.reshape(shape) for u in scaled.unbind(0)], alpha=-lr)

    def _adam_group_foreach(self, group) -> None:
        """`_adam_group` with multi-tensor launches (one rank, parameters on the GPU): the same arithmetic in the same order per
        element -- ~10 launches for the group instead of ~10 per parameter (82 parameters at Swift-B)."""
        decay = 1.0 - group["lr"] * group["weight_decay"]
        b1, b2 = group["betas"]
        ps = list(group["params"])
        for p in ps:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            st = self.state[p]
            if not st:
                st["exp_avg"], st["exp_avg_sq"], st["step"] = torch.zeros_like(p), torch.zeros_like(p), 0
            st["step"] += 1
        by_step = {}
        for p in ps:
            by_step.setdefault(self.state[p]["step"], []).append(p)
        for step, qs in by_step.items():
            grads = [p.grad for p in qs]
            m1, m2 = [self.state[p]["exp_avg"] for p in qs], [self.state[p]["exp_avg_sq"] for p in qs]
            torch._foreach_mul_(m1, b1)
            torch._foreach_add_(m1, grads, alpha=1.0 - b1)
            torch._foreach_mul_(m2, b2)
            torch._foreach_addcmul_(m2, grads, grads, value=1.0 - b2)
            den = torch._foreach_div(m2, 1.0 - b2 ** step)
            torch._foreach_sqrt_(den)
            torch._foreach_add_(den, group["eps"])
            upd = torch._foreach_div(m1, 1.0 - b1 ** step)
            torch._foreach_div_(upd, den)
            torch._foreach_mul_(qs, decay)
            torch._foreach_add_(qs, upd, alpha=-group["lr"])

    def _adam_group(self, group) -> None:
        decay = 1.0 - group["lr"] * group["weight_decay"]
        for p in group["params"]:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            st = self.state[p]
            if not st:
                st["exp_avg"], st["exp_avg_sq"], st["step"] = torch.zeros_like(p), torch.zeros_like(p), 0
            st["step"] += 1
            p.mul_(decay).add_(adam_update(p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"], group["betas"], group["eps"]),
                               alpha=-group["lr"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        multi = dist.is_available() and dist.is_initialized()  # (a one-rank group broadcasts to itself: same code path)
        world, rank = (dist.get_world_size(), dist.get_rank()) if multi else (1, 0)
        for group in self.param_groups:
            if group["use_muon"] and world == 1 and all(p.is_cuda for p in group["params"]):
                self._muon_group_stacked(group)  # nothing to exchange: same-shape matrices orthogonalised together
            elif group["use_muon"]:
                self._muon_group(group, world, rank, multi)
            elif world == 1 and all(p.is_cuda for p in group["params"]):
                self._adam_group_foreach(group)
            else:
                self._adam_group(group)
        return loss
