"""In-training validation rollout on the gfx950 path (mirrors reference src/swift/training/validate.py:23-127).

``RMSE_rollout(sampler, dataloader, dataset, target_interval, device, rng, num_batches)`` has the reference's signature
and returns its two values: the aggregate RMSE (sum over the 6 h / end-of-day checkpoints of the all-element RMSE) and the
``[n_vars, days + 1]`` array of latitude-weighted per-variable RMSEs.  The autoregressive state stays on the device:
forcings enter the network as a second condition tensor (no concat), the residual update and re-standardisation are
``swiftk_rollout_update`` (which also yields the physical-unit prediction) and the squared-error sums are
``swiftk_rmse_sums``; one device->host read of ``1 + n_vars`` floats per checkpoint.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from .. import ops
from .._lib import check, lib
from ..rollout import update_stats


def _lat_weights(dataset, H: int, device) -> torch.Tensor:
    lat, _ = dataset.get_lat_lon()
    w = torch.cos(torch.deg2rad(torch.as_tensor(np.asarray(lat), dtype=torch.float32)))
    return (w / w.mean()).to(device).contiguous()


@torch.no_grad()
def RMSE_rollout(sampler: Callable[..., torch.Tensor], dataloader, dataset, target_interval: int, device: torch.device,
                 rng: Optional[torch.Generator] = None, num_batches: Optional[int] = None, pipeline_engine: bool = False):
    per_day = 4
    nv = dataset.n_target_channels
    agg = 0.0
    sep = np.zeros([nv, target_interval // per_day + 1])
    if num_batches is None:
        num_batches = len(dataloader)
    # validate.py:88-96, 112-116 -- one helper with the generate rollout, so that the non-residual branch (state targets; SST
    # zeroed by zero_field) cannot drift apart between the two loops
    mx, sx, st = update_stats(dataset, 6, device)
    w_lat = None
    for _ in range(num_batches):
        X, TS, idx = next(dataloader)
        X = X.to(device, non_blocking=True).float().contiguous()       # B c h w, standardised
        TS = TS.to(device, non_blocking=True).float().contiguous()     # B days+1 c h w, physical units
        B, _, H, W = X.shape
        if w_lat is None:
            w_lat = _lat_weights(dataset, H, device)
        phys = torch.empty_like(X)
        sq = ops.zeros_acc(1 + nv, device=device)
        for i in range(target_interval):
            forc = dataset.standardize_x(torch.stack([dataset.get_forcings(int(j) + i) for j in idx], 0)).to(device).float()
            Y = sampler((X, forc.contiguous()), generator=rng)
            ops.rollout_update(X, Y, mx, sx, st, phys=phys)           # residual: phys = unstd(X) + unstd_t(Y); X <- std(phys)
            if (i + 1) % per_day == 0 or i == 0:
                day = (i + 1) // per_day
                ops.zero_acc_(sq)  # (swiftk_rmse_sums accumulates atomically)
                tgt = TS[:, day]
                check(lib().swiftk_rmse_sums(phys.data_ptr(), tgt.data_ptr(), TS.stride(0), w_lat.data_ptr(), sq.data_ptr(), B, nv,
                                             H, W, torch.cuda.current_stream().cuda_stream), "swiftk_rmse_sums")
                s = sq.cpu().double().numpy()
                agg += float(np.sqrt(s[0] / (B * nv * H * W)))
                sep[:, day] += np.sqrt(s[1:] / (B * H * W))
    return agg / num_batches, sep / num_batches
