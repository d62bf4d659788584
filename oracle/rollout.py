"""CPU restatement of the autoregressive ensemble rollout (TEST INFRASTRUCTURE).

Reference: generate.py:48-154 (loop), data/era5.py:110-184 (standardisation).
"""
from __future__ import annotations

from typing import Callable, Dict

import torch


class Stats:
    """Per-channel affine statistics with the reference's channel-count dispatch.

    data/era5.py:110-133: the stats vector covers variables+forcings; a tensor
    with ``len(variables)`` channels uses the leading slice, one with
    ``len(forcings)`` channels the trailing slice, anything else all of it.
    Residual targets: mean 0 and a per-interval std (:95-100).
    """

    def __init__(self, x_mean, x_std, t_std: Dict[int, torch.Tensor], n_vars: int, n_forc: int):
        self.x_mean = torch.as_tensor(x_mean, dtype=torch.float32).reshape(-1, 1, 1)
        self.x_std = torch.as_tensor(x_std, dtype=torch.float32).reshape(-1, 1, 1)
        self.t_std = {k: torch.as_tensor(v, dtype=torch.float32).reshape(-1, 1, 1) for k, v in t_std.items()}
        self.n_vars, self.n_forc = n_vars, n_forc

    def _sel(self, v, m, s):
        c = v.shape[1 if v.ndim == 4 else 0]
        if c == self.n_vars:
            return m[: self.n_vars], s[: self.n_vars]
        if c == self.n_forc:
            return m[self.n_vars:], s[self.n_vars:]
        return m, s

    def standardize_x(self, x, delta: int = 6):
        m, s = self._sel(x, self.x_mean, self.x_std)
        return (x - m) / s

    def unstandardize_x(self, x, delta: int = 6):
        m, s = self._sel(x, self.x_mean, self.x_std)
        return x * s + m

    def unstandardize_t(self, t, delta: int = 6):
        s = self.t_std[delta]
        return t * s + torch.zeros_like(s)


@torch.no_grad()
def rollout(sampler: Callable, stats: Stats, X0: torch.Tensor, forcings: Callable[[int], torch.Tensor],
            steps: int, interval: int = 6, residual: bool = True) -> torch.Tensor:
    """One member-batch of generate.py:85-131.

    ``X0`` [B, n_vars, H, W] standardised initial state; ``forcings(i)`` returns
    the *physical* forcing fields [B, n_forc, H, W] for lead step i (the
    reference reads file ``j + i*interval//6`` per sample, :101-112);
    ``sampler(cond) -> Y`` is the closure of generating/factory.py.  Returns the
    physical-unit trajectory [B, steps+1, n_vars, H, W].
    """
    nv = stats.n_vars
    X = X0
    out = [stats.unstandardize_x(X)]
    for i in range(steps):
        cond = torch.cat([X, stats.standardize_x(forcings(i))], dim=1)
        Y = sampler(cond)
        if residual:
            X_un = stats.unstandardize_x(cond)[:, :nv]
            X_phys = X_un + stats.unstandardize_t(Y, delta=int(interval))
            out.append(X_phys)
            X = stats.standardize_x(X_phys)
        else:
            out.append(stats.unstandardize_x(Y))
            X = Y
    return torch.stack(out, dim=1)
