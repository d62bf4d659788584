"""CPU restatement of the in-training validation rollout (TEST INFRASTRUCTURE ONLY).

Reference: training/validate.py:23-127 (``RMSE_rollout``): autoregressive rollout of one batch to ``target_interval``
six-hour steps, RMSE against unstandardised targets after the first step and at the end of every day (4 steps):
  * aggregate: sum over those checkpoints of sqrt(mean((Y - T)^2)) over everything;
  * per variable and checkpoint: sqrt(mean_{b,h,w}(w_lat (Y - T)^2)), w_lat = cos(lat) / mean(cos(lat)).
Pinned by tests/golden/val_tiny.npz (the reference function itself, run on the fake dataset).
"""
from __future__ import annotations

from typing import Callable

import numpy as np
import torch

from .rollout import Stats


@torch.no_grad()
def rmse_rollout(sampler: Callable, stats: Stats, X0, TS, forcings: Callable[[int], torch.Tensor], lat_deg,
                 target_interval: int, residual: bool = True):
    """X0 [B, nv, H, W] standardised; TS [B, days + 1, nv, H, W] physical targets (6 h, day 1, day 2, ...);
    ``forcings(i)`` physical forcings for step i; ``sampler(cond) -> Y``.  Returns (aggregate, [nv, days + 1])."""
    nv = stats.n_vars
    per_day = 4
    w_lat = torch.cos(torch.deg2rad(torch.as_tensor(lat_deg, dtype=torch.float32)))
    w_lat = (w_lat / w_lat.mean())[None, None, :, None]
    agg = 0.0
    sep = np.zeros([nv, target_interval // per_day + 1])
    X = X0
    for i in range(target_interval):
        Xc = torch.cat([X, stats.standardize_x(forcings(i))], dim=1)
        Y = sampler(Xc)
        if (i + 1) % per_day == 0 or i == 0:
            day = (i + 1) // per_day
            Y_un = stats.unstandardize_t(Y)
            if residual:
                Y_un = stats.unstandardize_x(Xc)[:, :nv] + Y_un
            T_un = TS[:, day]
            agg += float(torch.sqrt(torch.mean((Y_un - T_un) ** 2)))
            sep[:, day] += torch.sqrt(torch.mean(w_lat * ((Y_un - T_un) ** 2), dim=(0, 2, 3))).numpy()
        if residual:
            X = stats.standardize_x(stats.unstandardize_x(Xc)[:, :nv] + stats.unstandardize_t(Y))
        else:
            X = Y
    return agg, sep
