"""Offline ensemble metrics on the device (mirrors reference src/swift/eval/metrics.py:39-134).

``lat_weighted_rmse`` / ``lat_weighted_crps`` / ``lat_weighted_spread_skill_ratio`` keep the reference's signatures and key
names (``rmse_<var>_<postfix>`` ...); all three come out of ONE pass over the ensemble (``swiftk_ensemble_sums``: the N member
values of a grid point are read once into registers; the reference materialises a [B, N, N, H, W] difference tensor per
variable).  ``python -m swift_amd.eval.metrics --pred output.npy --truth truth.npy --lat lat.npy`` evaluates the
``--dump numpy`` output of ``swift_amd.generate`` (zarr / xarray are not available in this image).
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, Sequence

import numpy as np
import torch

from .._lib import check, lib


def ensemble_sums(pred: torch.Tensor, y: torch.Tensor, lat) -> torch.Tensor:
    """pred [B, N, V, H, W], y [B, V, H, W] (device, fp32) -> [B, V, 4] weighted grid sums (see include/swiftk.h)."""
    assert pred.is_cuda and y.is_cuda and pred.ndim == 5 and y.ndim == 4
    B, N, V, H, W = pred.shape
    w = np.cos(np.deg2rad(np.asarray(lat, dtype=np.float64)))
    w_lat = torch.from_numpy(w / w.mean()).float().to(pred.device).contiguous()
    pred, y = pred.contiguous().float(), y.contiguous().float()
    out = torch.zeros(B, V, 4, device=pred.device)
    check(lib().swiftk_ensemble_sums(pred.data_ptr(), y.data_ptr(), w_lat.data_ptr(), out.data_ptr(), B, N, V, H, W,
                                     torch.cuda.current_stream().cuda_stream), "swiftk_ensemble_sums")
    return out


def all_metrics(pred: torch.Tensor, y: torch.Tensor, vars: Sequence[str], lat, log_postfix: str) -> Dict[str, torch.Tensor]:
    B, N, V, H, W = pred.shape
    s = ensemble_sums(pred, y, lat).double()
    hw = H * W
    rmse = torch.sqrt(s[..., 0] / hw).mean(0)
    crps = s[..., 1].sum(0) / (B * N * hw) - (s[..., 2] / hw / (2 * N * (N - 1))).mean(0)
    ssr = torch.sqrt(s[..., 3] / hw).mean(0) / rmse
    out = {}
    for i, v in enumerate(vars):
        out[f"rmse_{v}_{log_postfix}"], out[f"crps_{v}_{log_postfix}"], out[f"ssr_{v}_{log_postfix}"] = rmse[i], crps[i], ssr[i]
    return out


def _pick(prefix):
    def fn(pred, y, vars, lat, log_postfix):
        if pred.ndim == 4:  # reference: deterministic input -> plain RMSE
            pred = torch.stack([pred, pred], 1)
        return {k: v for k, v in all_metrics(pred, y, vars, lat, log_postfix).items() if k.startswith(prefix)}
    return fn


lat_weighted_rmse = _pick("rmse_")
lat_weighted_crps = _pick("crps_")
lat_weighted_spread_skill_ratio = _pick("ssr_")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--pred", required=True, help="output-*.npy of swift_amd.generate: (samples, members, steps+1, C, H, W)")
    ap.add_argument("--truth", required=True, help="npy (samples, steps+1, C, H, W) in physical units")
    ap.add_argument("--lat", default=None, help="npy of latitudes (default: linspace(-90, 90, H))")
    ap.add_argument("--interval", type=int, default=6)
    a = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    pred, truth = np.load(a.pred, mmap_mode="r"), np.load(a.truth, mmap_mode="r")
    S, N, T, C, H, W = pred.shape
    lat = np.load(a.lat) if a.lat else np.linspace(-90, 90, H)
    names = [f"var{i}" for i in range(C)]
    res = {}
    for j in range(1, T):
        p = torch.from_numpy(np.ascontiguousarray(pred[:, :, j])).to(dev)
        y = torch.from_numpy(np.ascontiguousarray(truth[:, j])).to(dev)
        res.update({k: float(v) for k, v in all_metrics(p, y, names, lat, f"{j * a.interval}h").items()})
    path = os.path.join(os.path.dirname(os.path.abspath(a.pred)), "evaluation_metrics.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(f"wrote {path} ({len(res)} metrics)")


if __name__ == "__main__":
    main()
