"""Preconditioner wrapper that owns the network (mirrors reference src/swift/models/precond.py:101-151).

``PassPrecond`` = identity scaling + channel-concat of the condition.  The concat of
precond.py:139-141 is not materialised: the sources go to the patch-gather kernel as separate
pointers.  ``EDMPrecond`` (``*-edm`` experiments only) is out of scope.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..config import instantiate
from .abstract import AbstractNetwork


def _2d_resolution(x) -> np.ndarray:
    """An int or an (H, W) pair -> ``np.array([H, W])``, the type callers of the reference read from ``net.img_resolution``
    (precond.py:9-18)."""
    hw = np.broadcast_to(np.asarray(x, dtype=int).reshape(-1), (2,)) if np.ndim(x) == 0 else np.asarray(x, dtype=int)
    if hw.shape != (2,):
        raise AssertionError(f"img_resolution must be an int or a pair, got {x!r}")
    return np.array(hw)


def _process_auxiliary(auxiliary, auxiliary_dim, batch_size, device):
    """scalar / [1] / [B] / None -> [B|1, auxiliary_dim]  (precond.py:21-31).  None stands for a zero lead time (one row that the
    network broadcasts); a Python number is materialised by a fill kernel rather than a pageable host-to-device copy (capturable
    into a HIP graph); tensors on another device are brought over without a host stall."""
    if not auxiliary_dim:
        return None
    dev = torch.device(device)
    if auxiliary is None:
        return torch.zeros(1, auxiliary_dim, device=dev)
    if isinstance(auxiliary, (int, float)):
        aux = torch.full((), float(auxiliary), dtype=torch.float32, device=dev)
    else:
        aux = auxiliary if torch.is_tensor(auxiliary) else torch.as_tensor(auxiliary)
        if aux.device != dev:  # (the trainer keeps lead times on the host; pinned loader batches copy asynchronously)
            aux = aux.to(dev, non_blocking=True)
    if aux.numel() == 1 and aux.dim() <= 1:  # one value for the whole batch
        aux = aux.reshape(1).expand(batch_size)
    return aux.reshape(-1, auxiliary_dim)


class PassPrecond(torch.nn.Module):
    def __init__(
        self,
        model_config,
        img_resolution,
        img_channels: int,
        condition_channels: int = 0,
        auxiliary_dim: int = 0,
        sigma_min: float = 0.0,
        sigma_max: float = float("inf"),
        sigma_data: float = 1.0,
    ):
        super().__init__()
        self.img_resolution = _2d_resolution(img_resolution)
        self.img_channels = img_channels
        self.condition_channels = condition_channels
        self.auxiliary_dim = auxiliary_dim
        self.sigma_min, self.sigma_max, self.sigma_data = sigma_min, sigma_max, sigma_data
        self.model_config = model_config
        self.model: AbstractNetwork = instantiate(
            model_config,
            img_resolution=[int(v) for v in self.img_resolution],
            in_channels=img_channels + condition_channels,
            out_channels=img_channels,
            auxiliary_dim=auxiliary_dim,
            _convert_="object",
        )

    def forward(self, x, t, condition=None, auxiliary=None, **model_kwargs):
        """x [B,C,H,W], t [B], condition [B,Cc,H,W] -> F [B,C,H,W]  (precond.py:133-148).

        Extra keyword arguments beyond the reference's ``jvp`` / ``return_logvar``:
        ``x_scale`` (multiplies x inside the patch gather: x_t / sigma_d) and ``xt, alpha, beta``
        (out = alpha*xt + beta*F fused into the un-patchify) used by the samplers.
        """
        aux = _process_auxiliary(auxiliary, self.auxiliary_dim, x.size(0), x.device)
        model_kwargs.pop("jvp", None)
        x_scale = model_kwargs.pop("x_scale", 1.0)
        srcs, scales = [x], [x_scale]
        if condition is not None and self.condition_channels > 0:
            # additive extension: a (state, forcings) pair is accepted as-is, saving the caller's concat
            parts = list(condition) if isinstance(condition, (tuple, list)) else [condition]
            assert sum(p.shape[1] for p in parts) == self.condition_channels
            srcs += parts
            scales += [1.0] * len(parts)
        return self.model.forward_sources(srcs, scales, t.flatten(), aux, **model_kwargs)

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)
