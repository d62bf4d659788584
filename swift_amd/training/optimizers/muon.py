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
    """muon.py:5-35 for one 2-D matrix on the GPU; returns bf16 of G's shape."""
    assert G.ndim == 2 and G.is_cuda
    a, b, c = NS_COEFFS
    X = G.to(_BF)
    tall = G.size(0) > G.size(1)
    if tall:
        X = X.mT
    X = X / (X.norm(dim=(-2, -1), keepdim=True) + 1e-7)
    m, n = X.shape  # m <= n
    if m % 4 or n % 4 or m < 16:
        # degenerate "matrices" the reference's rule also sends through Muon (the [1, heads, 1, 1] logit scale viewed as
        # 1 x heads): a few dozen FLOPs, below the GEMM's shape granularity -> plain rocBLAS matmul
        for _ in range(steps):
            A = X @ X.mT
            X = a * X + (b * A + c * A @ A) @ X
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


def muon_update(grad: torch.Tensor, momentum: torch.Tensor, beta: float = 0.95, ns_steps: int = 5, nesterov: bool = True):
    """muon.py:38-45 (grad is consumed: the Nesterov blend is written into it, as the reference does)."""
    momentum.lerp_(grad, 1 - beta)
    update = grad.lerp_(momentum, beta) if nesterov else momentum
    if update.ndim == 4:
        update = update.view(len(update), -1)
    update = zeropower_via_newtonschulz5(update, steps=ns_steps)
    update = update * max(1, grad.size(-2) / grad.size(-1)) ** 0.5
    return update


def adam_update(grad, buf1, buf2, step, betas, eps):
    """muon.py:149-154."""
    buf1.lerp_(grad, 1 - betas[0])
    buf2.lerp_(grad.square(), 1 - betas[1])
    buf1c = buf1 / (1 - betas[0] ** step)
    buf2c = buf2 / (1 - betas[1] ** step)
    return buf1c / (buf2c.sqrt() + eps)


class MuonWithAuxAdam(torch.optim.Optimizer):
    def __init__(self, param_groups, **kwargs):
        param_groups = [dict(g) for g in param_groups]
        for group in param_groups:
            assert "use_muon" in group
            if group["use_muon"]:
                group["params"] = sorted(group["params"], key=lambda x: x.size(), reverse=True)
                group["lr"] = group.get("lr", 0.02)
                group["momentum"] = group.get("momentum", 0.95)
                group["weight_decay"] = group.get("weight_decay", 0)
                assert set(group.keys()) == {"params", "lr", "momentum", "weight_decay", "use_muon"}
            else:
                group["lr"] = group.get("lr", 3e-4)
                group["betas"] = group.get("betas", (0.9, 0.95))
                group["eps"] = group.get("eps", 1e-10)
                group["weight_decay"] = group.get("weight_decay", 0)
                assert set(group.keys()) == {"params", "lr", "betas", "eps", "weight_decay", "use_muon"}
        super().__init__(param_groups, dict())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        multi = dist.is_available() and dist.is_initialized()  # (a one-rank group broadcasts to itself: same code path)
        world, rank = (dist.get_world_size(), dist.get_rank()) if multi else (1, 0)
        for group in self.param_groups:
            if group["use_muon"]:
                params = group["params"]
                for base in range(0, len(params), world):
                    if base + rank < len(params):
                        p = params[base + rank]
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                        state = self.state[p]
                        if len(state) == 0:
                            state["momentum_buffer"] = torch.zeros_like(p)
                        update = muon_update(p.grad, state["momentum_buffer"], beta=group["momentum"])
                        p.mul_(1 - group["lr"] * group["weight_decay"])
                        p.add_(update.reshape(p.shape).to(p.dtype), alpha=-group["lr"])
                    if multi:
                        # the reference all-gathers the round's parameters (muon.py:234-237), which needs one shape per
                        # round; rounds here may mix shapes (12 x w1 then 12 x to_qkv on 8 ranks), so every parameter of
                        # the round is broadcast from the rank that owns it
                        for r in range(world):
                            if base + r < len(params):
                                dist.broadcast(params[base + r].data, src=r)
            else:
                for p in group["params"]:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                    state = self.state[p]
                    if len(state) == 0:
                        state["exp_avg"] = torch.zeros_like(p)
                        state["exp_avg_sq"] = torch.zeros_like(p)
                        state["step"] = 0
                    state["step"] += 1
                    update = adam_update(p.grad, state["exp_avg"], state["exp_avg_sq"], state["step"], group["betas"], group["eps"])
                    p.mul_(1 - group["lr"] * group["weight_decay"])
                    p.add_(update, alpha=-group["lr"])
        return loss
