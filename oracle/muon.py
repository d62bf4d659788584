"""CPU restatement of the reference's MuonWithAuxAdam update rule (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows reference src/swift/training/optimizers/muon.py: ``zeropower_via_newtonschulz5`` (:5-35), ``muon_update`` (:38-45),
``adam_update`` (:149-154) and the per-parameter step of ``(SingleDevice)MuonWithAuxAdam`` (:157-264, :267-338).
Pinned by tests/golden/muon_tiny.npz, produced by running the reference's SingleDeviceMuonWithAuxAdam (tools/make_golden.py).
"""
from __future__ import annotations

import torch

NS_COEFFS = (3.4445, -4.7750, 2.0315)


def newton_schulz5(G: torch.Tensor, steps: int = 5) -> torch.Tensor:
    """Quintic Newton-Schulz orthogonalisation in bf16 (muon.py:5-35): X <- aX + (bA + cA^2) X, A = X X^T."""
    a, b, c = NS_COEFFS
    X = G.bfloat16()
    tall = G.size(-2) > G.size(-1)
    if tall:
        X = X.mT
    X = X / (X.norm(dim=(-2, -1), keepdim=True) + 1e-7)
    for _ in range(steps):
        A = X @ X.mT
        B = b * A + c * A @ A
        X = a * X + B @ X
    return X.mT if tall else X


def muon_update(grad: torch.Tensor, momentum: torch.Tensor, beta: float = 0.95, ns_steps: int = 5) -> torch.Tensor:
    """Nesterov momentum, orthogonalise, rescale by sqrt(max(1, rows/cols)) (muon.py:38-45).  Updates `momentum` in place."""
    momentum.lerp_(grad, 1 - beta)
    upd = grad.lerp(momentum, beta)
    upd = newton_schulz5(upd, ns_steps)
    return upd * max(1, grad.size(-2) / grad.size(-1)) ** 0.5


def adam_update(grad, buf1, buf2, step: int, betas, eps: float) -> torch.Tensor:
    """Bias-corrected Adam direction (muon.py:149-154).  Updates the moment buffers in place."""
    buf1.lerp_(grad, 1 - betas[0])
    buf2.lerp_(grad.square(), 1 - betas[1])
    return (buf1 / (1 - betas[0] ** step)) / ((buf2 / (1 - betas[1] ** step)).sqrt() + eps)


def step(params, grads, state, groups):
    """One optimiser step over `groups` = [dict(idx=[...], use_muon, lr, weight_decay, momentum | betas, eps)]; in place."""
    for g in groups:
        for i in g["idx"]:
            p, gr = params[i], grads[i]
            st = state.setdefault(i, {})
            if g["use_muon"]:
                if not st:
                    st["momentum_buffer"] = torch.zeros_like(p)
                upd = muon_update(gr.clone(), st["momentum_buffer"], beta=g.get("momentum", 0.95)).to(p.dtype)
            else:
                if not st:
                    st.update(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), step=0)
                st["step"] += 1
                upd = adam_update(gr, st["exp_avg"], st["exp_avg_sq"], st["step"], g["betas"], g["eps"])
            p.mul_(1 - g["lr"] * g["weight_decay"])
            p.add_(upd.reshape(p.shape), alpha=-g["lr"])
