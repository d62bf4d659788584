"""TrigFlow / sCM samplers on the gfx950 engine (mirrors reference src/swift/generating/diffusion.py).

Same solver names, keyword arguments and time grids as the reference's ``DiffusionSampler``
(scm_solver :417-461, dpm_solver_2s :355-415).  Differences in *how*, not *what*:
  * ``x_t / sigma_d`` is folded into the patch-gather kernel and the state update
    ``cos(t) x_t - sin(t) sigma_d F`` (or the Euler step) into the un-patchify epilogue, so a
    1-step sCM sample is exactly one fused network call;
  * the remaining state arithmetic (re-noising, Heun average) runs in ``swiftk_axpby``;
  * the time grid is built on the host in fp32 with the same torch ops as the reference and
    read back once (no per-step device->host sync).
``dpm_solver`` (:289-353, the in-training validation solver) is built the same way.  EDM-family solvers (edm_sampler,
ablation_sampler, scm_solve2) are out of scope.
"""
from __future__ import annotations

import contextlib
import math
from typing import Callable, Optional, Sequence

import torch

from .. import ops


@contextlib.contextmanager
def engine_for(module, dtype, device_type: str = "cuda"):
    """Select the compute engine of ``denoise_dtype`` for the calls inside the block: torch.bfloat16 = the bf16 MFMA engine
    (through autocast, as the reference's ``autocast(dtype=denoise_dtype)``, diffusion.py:457), torch.float32 = the
    exact-fp32 engine, the string "bf16x3" = the fp32-grade engine built from three bf16 products per GEMM (fp32
    activations; what ``generate --dtype bf16x3`` runs)."""
    model = getattr(module, "model", module)
    x3 = isinstance(dtype, str) and dtype == "bf16x3"
    if isinstance(dtype, str) and not x3:
        raise ValueError(f"denoise_dtype {dtype!r}: expected torch.float32, torch.bfloat16 or 'bf16x3'")
    prev = getattr(model, "fp32_engine", None)
    if x3:
        model.fp32_engine = "bf16x3"
    try:
        with torch.autocast(device_type, enabled=(dtype == torch.bfloat16), dtype=torch.bfloat16):
            yield
    finally:
        if x3:
            model.fp32_engine = prev


def _log_sigma_grid(num_steps: int, sigma_min: float, sigma_max: float, sigma_data: float) -> torch.Tensor:
    lo, hi = torch.log(torch.tensor(sigma_min)), torch.log(torch.tensor(sigma_max))
    u = torch.linspace(1, 0, num_steps)
    return torch.atan(torch.exp(lo + u * (hi - lo)) / sigma_data)


def scm_time_steps(num_steps, sigma_min, sigma_max, sigma_data, intermediates=None) -> torch.Tensor:
    """diffusion.py:434-449 (host tensor, fp32)."""
    if num_steps == 1:
        t = torch.tensor([torch.pi / 2])
    else:
        t = _log_sigma_grid(num_steps, sigma_min, sigma_max, sigma_data)
    t = torch.cat([t, torch.zeros(1)])
    if num_steps == 2 and intermediates is None:
        t = torch.tensor([t[0], 1.1, 0.0])
    elif intermediates:
        t = torch.cat([t[:1], torch.as_tensor(intermediates, dtype=torch.float32), t[-1:]])
    return t


class DiffusionSampler:
    def __init__(self, net):
        self.net = net

    def _module(self):
        return getattr(self.net, "module", self.net)  # net may be DDP (diffusion.py:433)

    def _call(self, x_t, t_scalar: torch.Tensor, condition, auxiliary, alpha: float, beta: float, dtype):
        """alpha * x_t + beta * net(x_t / sigma_d, t, condition, auxiliary) in one fused launch sequence."""
        B = x_t.shape[0]
        dev = x_t.device
        sd = self._module().sigma_data
        tt = torch.full((B,), float(t_scalar), dtype=torch.float32, device=dev)
        a = torch.full((B,), float(alpha), dtype=torch.float32, device=dev)
        b = torch.full((B,), float(beta), dtype=torch.float32, device=dev)
        with engine_for(self._module(), dtype, dev.type):
            return self.net(x_t, tt, condition, auxiliary, x_scale=1.0 / sd, xt=x_t, alpha=a, beta=b)

    @torch.no_grad()
    def scm_solver(self, latents: torch.Tensor, condition=None, auxiliary=None, randn_like: Callable = torch.randn_like,
                   num_steps: int = 2, intermediates: Optional[Sequence[float]] = None, sigma_min: float = 0.002,
                   sigma_max: float = 80.0, denoise_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """Multistep consistency sampler (diffusion.py:417-461)."""
        sd = self._module().sigma_data
        ts = scm_time_steps(num_steps, sigma_min, sigma_max, sd, intermediates)
        cos, sin = torch.cos(ts), torch.sin(ts)  # fp32, as the reference's device-side torch.cos/sin
        x_t = latents.contiguous().float() if sd == 1.0 else (latents * sd).contiguous().float()
        for i in range(len(ts) - 1):
            if i > 0:
                noise = randn_like(x_t).contiguous()
                x_t = ops.axpby(float(sin[i]) * sd, noise, float(cos[i]), x_t)
            x_t = self._call(x_t, ts[i], condition, auxiliary, float(cos[i]), -float(sin[i]) * sd, denoise_dtype)
        return x_t

    @torch.no_grad()
    def dpm_solver_2s(self, latents: torch.Tensor, condition=None, auxiliary=None, randn_like: Callable = torch.randn_like,
                      num_steps: int = 20, sigma_min: float = 0.002, sigma_max: float = 80.0, S_churn: float = 0.0,
                      S_min: float = 0.0, S_max: float = 1.57, S_noise: float = 1.0,
                      denoise_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """DPM-Solver++ 2S / Heun on the TrigFlow ODE (diffusion.py:355-415): 2*num_steps-1 network calls."""
        sd = self._module().sigma_data
        ts = torch.cat([_log_sigma_grid(num_steps, sigma_min, sigma_max, sd), torch.zeros(1)])
        x_t = (latents * sd).contiguous().float()
        for k in range(num_steps):
            s, t = ts[k], ts[k + 1]
            dt = float((t - s) * sd)  # fp32 (t - s), then * sigma_d as in "delta * sigma_data * F"
            x_e = self._call(x_t, s, condition, auxiliary, 1.0, dt, denoise_dtype)  # x + delta*sd*F_s
            if k < num_steps - 1:
                # x + delta*sd*0.5*(F_s + F_t) = 0.5*(x + x_e) + 0.5*delta*sd*F_t, with F_t evaluated at x_e
                half = ops.axpby(0.5, x_t, 0.5, x_e)
                B, dev = x_t.shape[0], x_t.device
                tt = torch.full((B,), float(t), dtype=torch.float32, device=dev)
                a = torch.ones(B, dtype=torch.float32, device=dev)
                b = torch.full((B,), 0.5 * dt, dtype=torch.float32, device=dev)
                with engine_for(self._module(), denoise_dtype, dev.type):
                    x_t = self.net(x_e, tt, condition, auxiliary, x_scale=1.0 / sd, xt=half, alpha=a, beta=b)
            else:
                x_t = x_e
        return x_t

    @torch.no_grad()
    def dpm_solver(self, latents: torch.Tensor, condition=None, auxiliary=None, randn_like: Callable = torch.randn_like,
                   num_steps: int = 20, use_pp: bool = True, sigma_min: float = 0.002, sigma_max: float = 80.0, rho: float = 7,
                   denoise_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """DPM-Solver(++) multistep on the TrigFlow ODE, EDM rho grid (diffusion.py:289-353): num_steps network calls; the
        first and last steps are first-order (DDIM), the others add the 2nd-order correction from the previous prediction."""
        sd = self._module().sigma_data
        ramp = torch.linspace(0, 1, num_steps)
        lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
        ts = torch.atan((hi + ramp * (lo - hi)) ** rho / sd)
        ts = torch.cat([ts, torch.zeros_like(ts[:1])])
        logtan = lambda u: torch.log(torch.tan(torch.clamp(u, 1e-4, 1.569)))
        x_t = (latents * sd).contiguous().float()
        t_prev = pred_prev = None
        for k in range(num_steps):
            s, t = ts[k], ts[k + 1]
            delta = s - t
            F = self._call(x_t, s, condition, auxiliary, 0.0, 1.0, denoise_dtype)        # F_s itself
            nxt = ops.axpby(float(torch.cos(delta)), x_t, -float(torch.sin(delta)) * sd, F)
            if use_pp:
                pred, denom = ops.axpby(float(torch.cos(s)), x_t, -float(torch.sin(s)) * sd, F), float(torch.sin(s))
            else:
                pred, denom = ops.axpby(float(torch.sin(s)), x_t, float(torch.cos(s)) * sd, F), float(torch.cos(s))
            if not (k == 0 or k == num_steps - 1):
                r_s = (logtan(s) - logtan(t_prev)) / (logtan(s) - logtan(t))
                coef = float(torch.sin(delta) / (2 * r_s * max(denom, 1e-3)))
                corr = ops.axpby(coef, pred_prev, -coef, pred)
                nxt = ops.axpby(1.0, nxt, 1.0 if use_pp else -1.0, corr)
            t_prev, pred_prev, x_t = s, pred, nxt
        return x_t
