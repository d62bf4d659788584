"""CPU restatement of the TrigFlow / sCM samplers (TEST INFRASTRUCTURE).

Reference: generating/diffusion.py:355-461, generating/factory.py:8-97.
``net`` is any PassPrecond-shaped callable ``net(x, t, condition, auxiliary)``
exposing ``sigma_data``.  Noise is an explicit argument (``latents`` and a
``renoise`` list) so that parity tests do not depend on an RNG stream.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch


def log_sigma_grid(num_steps: int, sigma_min: float, sigma_max: float, sigma_data: float, device=None):
    """t_k = atan(exp(lerp(ln smax -> ln smin)) / sigma_d)  (diffusion.py:378-383, 437-441)."""
    lo = torch.log(torch.tensor(sigma_min, device=device))
    hi = torch.log(torch.tensor(sigma_max, device=device))
    u = torch.linspace(1, 0, num_steps, device=device)
    return torch.atan(torch.exp(lo + u * (hi - lo)) / sigma_data)


def scm_time_steps(num_steps: int, sigma_min: float, sigma_max: float, sigma_data: float,
                   intermediates: Optional[Sequence[float]] = None, device=None) -> torch.Tensor:
    """diffusion.py:434-449."""
    if num_steps == 1:
        t = torch.tensor([math.pi / 2], device=device)
    else:
        t = log_sigma_grid(num_steps, sigma_min, sigma_max, sigma_data, device)
    t = torch.cat([t, torch.zeros(1, device=device)])
    if num_steps == 2 and intermediates is None:
        t = torch.tensor([t[0], 1.1, 0.0], device=device)
    elif intermediates:
        t = torch.cat([t[:1], torch.as_tensor(intermediates, device=device), t[-1:]])
    return t


@torch.no_grad()
def scm_solver(net, latents, condition=None, auxiliary=None, renoise: Optional[Sequence[torch.Tensor]] = None,
               num_steps: int = 2, intermediates=None, sigma_min: float = 0.002, sigma_max: float = 80.0):
    """Multistep consistency sampler (diffusion.py:417-461).

    ``renoise[i-1]`` is the N(0,1) tensor the reference draws with
    ``randn_like`` before network call i >= 1.
    """
    sd = net.sigma_data
    B = latents.shape[0]
    ts = scm_time_steps(num_steps, sigma_min, sigma_max, sd, intermediates, latents.device)
    x = latents * sd
    for i, t in enumerate(ts[:-1]):
        if i > 0:
            x = torch.sin(t) * (sd * renoise[i - 1]) + torch.cos(t) * x
        Fx = net(x / sd, t.expand(B), condition, auxiliary)
        x = torch.cos(t) * x - torch.sin(t) * sd * Fx
    return x


@torch.no_grad()
def dpm_solver_2s(net, latents, condition=None, auxiliary=None, num_steps: int = 20,
                  sigma_min: float = 0.002, sigma_max: float = 80.0):
    """Heun 2S on the TrigFlow ODE (diffusion.py:355-415); 2*num_steps-1 net calls."""
    sd = net.sigma_data
    B = latents.size(0)
    ts = torch.cat([log_sigma_grid(num_steps, sigma_min, sigma_max, sd, latents.device),
                    torch.zeros(1, device=latents.device)])
    x = latents * sd
    for k in range(num_steps):
        s, t = ts[k], ts[k + 1]
        dt = t - s
        Fs = net(x / sd, s.repeat(B), condition, auxiliary)
        xe = x + dt * sd * Fs
        if k < num_steps - 1:
            Ft = net(xe / sd, t.repeat(B), condition, auxiliary)
            x = x + dt * sd * 0.5 * (Fs + Ft)
        else:
            x = xe
    return x


@torch.no_grad()
def dpm_solver(net, latents, condition=None, auxiliary=None, num_steps: int = 20, use_pp: bool = True,
               sigma_min: float = 0.002, sigma_max: float = 80.0, rho: float = 7.0):
    """DPM-Solver(++) on the TrigFlow ODE with the EDM rho time grid (diffusion.py:289-353): num_steps net calls,
    first and last step first-order (DDIM), 2nd-order multistep correction in between."""
    sd = net.sigma_data
    B = latents.size(0)
    ramp = torch.linspace(0, 1, num_steps, device=latents.device)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    sigmas = (hi + ramp * (lo - hi)) ** rho
    ts = torch.atan(sigmas / sd)
    ts = torch.cat([ts, torch.zeros_like(ts[:1])])
    logtan = lambda u: torch.log(torch.tan(torch.clamp(u, 1e-4, 1.569)))
    x = latents * sd
    t_prev = pred_prev = None
    for k in range(num_steps):
        s, t = ts[k], ts[k + 1]
        delta = s - t
        Fs = net(x / sd, s.repeat(B), condition, auxiliary)
        if use_pp:
            pred, denom = torch.cos(s) * x - torch.sin(s) * sd * Fs, torch.sin(s)
        else:
            pred, denom = torch.sin(s) * x + torch.cos(s) * sd * Fs, torch.cos(s)
        nxt = torch.cos(delta) * x - torch.sin(delta) * sd * Fs
        if not (k == 0 or k == num_steps - 1):
            r_s = (logtan(s) - logtan(t_prev)) / (logtan(s) - logtan(t))
            corr = (torch.sin(delta) / (2 * r_s * max(denom, 1e-3))) * (pred_prev - pred)
            nxt = nxt + (corr if use_pp else -corr)
        t_prev, pred_prev, x = s, pred, nxt
    return x
