"""Sampler closures (mirrors reference src/swift/generating/factory.py:8-97).

``sampler_factory(mode, net, denoise_dtype, **solver_kwargs) -> sampler(X, generator)`` with
modes "scm" and "2s".  Latents are drawn exactly like the reference (``torch.randn`` with the
caller's generator on X's device, factory.py:52-56) -- RNG is torch plumbing, not part of the
hand-written path; parity tests inject latents through ``latents=``.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .diffusion import DiffusionSampler


def sampler_factory(mode: str, net: torch.nn.Module, denoise_dtype: torch.dtype = torch.float32,
                    **solver_kwargs) -> Callable[..., torch.Tensor]:
    O = DiffusionSampler(net)
    mod = getattr(net, "module", net)
    if mode == "scm":
        solve = O.scm_solver
    elif mode == "2s":
        solve = O.dpm_solver_2s
    elif mode == "dpm":
        solve = O.dpm_solver
    elif mode == "edm":
        raise NotImplementedError(f"solver mode {mode!r} is outside the sCM/TrigFlow forecast path built here")
    else:
        raise ValueError(f"Unknown solver mode: {mode}")

    def sampler(X: torch.Tensor, generator: Optional[torch.Generator] = None, *args,
                latents: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        if latents is None:
            X0 = X[0] if isinstance(X, (tuple, list)) else X  # condition may arrive as (state, forcings), never concatenated
            latents = torch.randn((X0.shape[0], mod.img_channels, *mod.img_resolution), generator=generator,
                                  device=X0.device)
        return solve(latents=latents, condition=X, denoise_dtype=denoise_dtype, **solver_kwargs)

    return sampler
