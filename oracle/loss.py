"""CPU restatement of the TrigFlow / sCM / multistep-CRPS losses (TEST INFRASTRUCTURE).

Reference: training/loss.py.  All random draws are explicit arguments
(``tau`` [B,1,1,1], ``z`` N(0,1) like x, per-step latents for CRPS) so the
numbers can be compared without sharing an RNG stream.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import numpy as np
import torch

PRESSURE_LEVELS = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
LEVEL_VARS = ["geopotential", "u_component_of_wind", "v_component_of_wind", "vertical_velocity", "wind_speed",
              "temperature", "relative_humidity", "specific_humidity", "vorticity", "potential_vorticity"]
SURFACE_W = {"2m_temperature": 1.0, "sea_surface_temperature": 0.1, "10m_u_component_of_wind": 0.1,
             "10m_v_component_of_wind": 0.1, "mean_sea_level_pressure": 0.1}


def latitude_weights(lat_dim: int) -> torch.Tensor:
    """cos(lat)/mean, clamped >= 0.1, shape [1,1,H,1] (loss.py:28-32)."""
    w = torch.cos(torch.deg2rad(torch.linspace(-90, 90, lat_dim)))
    return torch.clamp(w / w.mean(), min=0.1).view(1, 1, -1, 1)


def variable_weights(variables: Sequence[str]) -> torch.Tensor:
    """Surface table + p/sum(p) per level, normalised to sum 1, [1,C,1,1] (loss.py:35-55)."""
    tot = sum(PRESSURE_LEVELS)
    table = dict(SURFACE_W)
    for v in LEVEL_VARS:
        for l in PRESSURE_LEVELS:
            table[f"{v}_{l}"] = l / tot
    w = torch.Tensor([table[v] for v in variables]).view(1, -1, 1, 1)
    return w / w.sum()


def loguniform_from_u(u: torch.Tensor, sigma_min: float, sigma_max: float) -> torch.Tensor:
    """loss.py:66-71 with the uniform draw ``u`` [B,1,1,1] given."""
    lo, hi = torch.log(torch.tensor(sigma_min)), torch.log(torch.tensor(sigma_max))
    return torch.exp(lo + u * (hi - lo))


def _split_out(out, x):
    if isinstance(out, tuple):
        return out[0], out[1].reshape(-1, 1, 1, 1)
    return out, torch.zeros_like(x[:, 0:1, 0:1, 0:1])


def trigflow_loss(net: Callable, x, tau, z, w_var, w_lat, sigma_data: float = 1.0, condition=None, auxiliary=None,
                  return_logvar: bool = False):
    """loss.py:132-160.  ``z`` is N(0,1); the reference scales it by sigma_data."""
    t = torch.atan(tau / sigma_data)
    z = z * sigma_data
    c, s = torch.cos(t), torch.sin(t)
    x_t = c * x + s * z
    v_t = c * z - s * x
    out = net(x_t / sigma_data, t, condition, auxiliary, return_logvar=return_logvar)
    Fx, lv = _split_out(out, x)
    return ((1 / torch.exp(lv)) * (w_var * w_lat * torch.square(sigma_data * Fx - v_t)) + lv).sum(dim=1).mean()


def scm_loss(net: Callable, x, tau, z, w_var, w_lat, step: int, sigma_data: float = 1.0, tangent_warmup_kimg: int = 0,
             condition=None, auxiliary=None, return_logvar: bool = False, teacher: Optional[Callable] = None):
    """loss.py:186-260 (continuous-time consistency loss, forward-mode tangent)."""
    t = torch.atan(tau / sigma_data)
    z = z * sigma_data
    c, s = torch.cos(t), torch.sin(t)
    x_t = c * x + s * z
    if teacher is not None:
        with torch.no_grad():
            dxt = sigma_data * teacher(x_t / sigma_data, t, condition, auxiliary)
    else:
        dxt = c * z - s * x

    def f(xx, tt):
        return net(xx, tt, condition, auxiliary, jvp=True)

    _, dF = torch.func.jvp(f, (x_t / sigma_data, t), (c * s * dxt / sigma_data, c * s))
    out = net(x_t / sigma_data, t, condition, auxiliary, return_logvar=return_logvar)
    Fx, lv = _split_out(out, x)
    r = min(1.0, step / (tangent_warmup_kimg * 1000)) if tangent_warmup_kimg > 0 else 1.0
    g = -(c ** 2) * (sigma_data * Fx.detach() - dxt) - r * ((c * s) * x_t + sigma_data * dF.detach())
    gn = torch.linalg.vector_norm(g, dim=(1, 2, 3), keepdim=True)
    gn = gn * np.sqrt(gn.numel() / g.numel())
    g = g / (gn + 0.1)
    return ((1 / torch.exp(lv)) * (w_var * w_lat * torch.square(Fx - Fx.detach() - g)) + lv).sum(dim=1).mean()


def almost_fair_crps(preds: torch.Tensor, target: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """preds [m, ...], target [...] -> [...]  (loss.py:343-371).

    mean_i |X_i - y|  -  (1 - (1-alpha)/m) / (2 m (m-1)) * sum_{i != j} |X_i - X_j|.
    """
    m = preds.shape[0]
    assert m > 1
    eps = (1.0 - alpha) / m
    skill = (preds - target.unsqueeze(0)).abs().mean(0)
    spread = (preds.unsqueeze(0) - preds.unsqueeze(1)).abs().sum(dim=(0, 1)) / (2 * m * (m - 1))
    return skill - (1 - eps) * spread


def crps_multistep_loss(net: Callable, stats, target, condition, auxiliary, forcings: Callable[[int], torch.Tensor],
                        latents: Sequence[Sequence[torch.Tensor]], w_var, w_lat, steps: int = 1,
                        sigma_data: float = 1.0, alpha: float = 1.0):
    """loss.py:373-445 without activation checkpointing (same numbers).

    ``latents[e][i]`` is the N(0,1) draw of ensemble member e at sub-step i;
    ``forcings(i)`` the physical forcing fields for sub-step i (the reference
    reads ``idx + i*dt*10//6`` per sample, :389-397).  ``delta`` = int(aux[0]*10).
    """
    nv = stats.n_vars
    B = target.shape[0]
    t = torch.tensor(math.pi / 2, dtype=target.dtype)
    delta = int(auxiliary[0] * 10)
    preds = []
    for e in range(len(latents)):
        cond = condition[:, :nv]
        pred = None
        for i in range(steps):
            x_t = latents[e][i] * sigma_data
            cc = torch.cat([cond, stats.standardize_x(forcings(i))], dim=1)
            out = net(x_t / sigma_data, t.expand(B), cc, auxiliary)
            pred = -sigma_data * out
            x_un = stats.unstandardize_x(cc, delta)[:, :nv]
            cond = stats.standardize_x(x_un + stats.unstandardize_t(pred, delta), delta)
        preds.append(pred)
    crps = almost_fair_crps(torch.stack(preds, 0), target, alpha)
    return (w_var * w_lat * crps).sum(dim=1).mean()
