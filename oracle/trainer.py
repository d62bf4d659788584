"""CPU restatement of the reference's optimisation step (src/swift/training/trainer.py:199-247) -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU-baseline leg may import this package (see
``oracle/__init__.py``).  Pinned by ``tests/golden/trainer_tiny.npz``, which ``tools/make_golden.py`` produced by calling
the reference's own ``Trainer._backward_step`` (AdamW is ``torch.optim.AdamW`` there; its update rule is restated here
from torch/optim/adamw.py: decoupled weight decay, bias-corrected moments, ``eps`` added to ``sqrt(v_hat)``).
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch


def learning_rate(global_nimg: int, base_lr: float, lr_rampup_kimg: float, lr_min_factor: float, lr_cosine_anneal: bool,
                  total_kimg: float, current: float) -> float:
    """trainer.py:201-217: linear warm-up from ``base * min_factor``, then cosine back down to it (or unchanged)."""
    warm = lr_rampup_kimg * 1000
    lo = base_lr * lr_min_factor
    if global_nimg < warm:
        return lo + (base_lr - lo) * (global_nimg / warm)
    if lr_cosine_anneal:
        prog = min(1.0, (global_nimg - warm) / (total_kimg * 1000 - warm))
        return lo + 0.5 * (base_lr - lo) * (1 + math.cos(math.pi * prog))
    return current


def sanitize(g: torch.Tensor) -> torch.Tensor:
    """trainer.py:223-231: nan -> 0, +inf -> 1e5, -inf -> -1e5."""
    g = torch.where(torch.isnan(g), torch.zeros_like(g), g)
    g = torch.where(g == float("inf"), torch.full_like(g, 1e5), g)
    return torch.where(g == float("-inf"), torch.full_like(g, -1e5), g)


def ema_beta(global_batch_size: int, ema_halflife_kimg: float, ema_rampup_ratio, global_nimg: int) -> float:
    """trainer.py:238-244."""
    half = ema_halflife_kimg * 1000
    if ema_rampup_ratio is not None:
        half = min(half, global_nimg * ema_rampup_ratio)
    return 0.5 ** (global_batch_size / max(half, 1e-8))


def adamw_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, betas, eps: float,
                 weight_decay: float):
    """One AdamW step on fp32 tensors (returns new p, m, v)."""
    b1, b2 = betas
    p = p * (1.0 - lr * weight_decay)
    m = m + (g - m) * (1.0 - b1)
    v = v * b2 + (1.0 - b2) * g * g
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


class OracleTrainerState:
    """Parameters, AdamW moments and EMA copies of a list of tensors, advanced by :meth:`step` exactly as
    ``Trainer._backward_step`` advances the reference's (gradients arrive as tensors)."""

    def __init__(self, params: Sequence[torch.Tensor], ema: Sequence[torch.Tensor], group_of: Sequence[int],
                 base_lr: Sequence[float], weight_decay: Sequence[float], betas=(0.9, 0.999), eps: float = 1e-8):
        self.p = [t.detach().clone().float() for t in params]
        self.e = [t.detach().clone().float() for t in ema]
        self.m = [torch.zeros_like(t) for t in self.p]
        self.v = [torch.zeros_like(t) for t in self.p]
        self.group_of, self.base_lr, self.wd = list(group_of), list(base_lr), list(weight_decay)
        self.lr = list(base_lr)
        self.betas, self.eps, self.t = betas, eps, 0

    def step(self, grads: List[torch.Tensor], global_nimg: int, *, lr_rampup_kimg, lr_min_factor, lr_cosine_anneal, total_kimg,
             global_batch_size, ema_halflife_kimg, ema_rampup_ratio):
        self.lr = [learning_rate(global_nimg, b, lr_rampup_kimg, lr_min_factor, lr_cosine_anneal, total_kimg, c)
                   for b, c in zip(self.base_lr, self.lr)]
        self.t += 1
        beta = ema_beta(global_batch_size, ema_halflife_kimg, ema_rampup_ratio, global_nimg)
        for i, g in enumerate(grads):
            k = self.group_of[i]
            self.p[i], self.m[i], self.v[i] = adamw_update(self.p[i], sanitize(g.float()), self.m[i], self.v[i], self.t,
                                                           self.lr[k], self.betas, self.eps, self.wd[k])
            self.e[i] = self.p[i] + beta * (self.e[i] - self.p[i])  # p_net.lerp(p_ema, beta)  (trainer.py:245-246)
        return self.lr
