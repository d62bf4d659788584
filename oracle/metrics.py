"""CPU restatement of the offline ensemble metrics (TEST INFRASTRUCTURE ONLY).

Reference: eval/metrics.py:39-134 -- latitude-weighted ensemble-mean RMSE, CRPS and spread/skill ratio per variable.
pred [B, N, V, H, W] (N members), y [B, V, H, W]; w_lat = cos(lat) / mean(cos(lat)).  Pinned by tests/golden/metrics_tiny.npz
(the reference functions themselves).
"""
from __future__ import annotations

import numpy as np
import torch


def _w(lat, like):
    w = np.cos(np.deg2rad(np.asarray(lat)))
    return torch.from_numpy(w / w.mean()).to(like.dtype)


def rmse(pred, y, lat):
    """[V]: mean_b sqrt(mean_{h,w} w (ens_mean - y)^2)   (:39-66)."""
    if pred.ndim == 5:
        pred = pred.mean(dim=1)
    w = _w(lat, y).view(1, 1, -1, 1)
    return torch.sqrt(((pred - y) ** 2 * w).mean(dim=(-2, -1))).mean(dim=0)


def crps(pred, y, lat):
    """[V]: mean w |x_n - y|  -  mean_b sum_{n,n'} mean_{h,w} w |x_n - x_n'| / (2 N (N-1))   (:69-106)."""
    N = pred.shape[1]
    w = _w(lat, y).view(1, 1, 1, -1, 1)
    err = ((pred - y.unsqueeze(1)).abs() * w).mean(dim=(0, 1, 3, 4))
    spread = ((pred.unsqueeze(2) - pred.unsqueeze(1)).abs() * w.unsqueeze(0)).mean(dim=(-2, -1)).sum(dim=(1, 2)) / (2 * N * (N - 1))
    return err - spread.mean(dim=0)


def spread_skill_ratio(pred, y, lat):
    """[V]: mean_b sqrt(mean_{h,w} w var_n(x)) / rmse   (:109-134; torch.var is the unbiased estimator)."""
    w = _w(lat, y).view(1, 1, -1, 1)
    spread = (torch.var(pred, dim=1) * w).mean(dim=(-2, -1)).sqrt().mean(dim=0)
    return spread / rmse(pred, y, lat)
