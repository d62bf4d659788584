"""Offline ensemble metrics on the device (mirrors reference src/swift/eval/metrics.py:39-134).

``lat_weighted_rmse`` / ``lat_weighted_crps`` / ``lat_weighted_spread_skill_ratio`` keep the reference's signatures and key
names (``rmse_<var>_<postfix>`` ...); all three come out of ONE pass over the ensemble (``swiftk_ensemble_sums``: the N member
values of a grid point are read once into registers; the reference materialises a [B, N, N, H, W] difference tensor per
variable).  ``python -m swift_amd.eval.metrics --truth T.zarr --pred P.zarr`` is the reference's CLI (eval/metrics.py:157-280: both
stores, the structured evaluation_metrics.json), walking the forecast store one initial condition at a time through the
standard-library zarr reader (zarr / xarray are not in this image); ``--pred output.npy --truth truth.npy --lat lat.npy`` evaluates
the ``--dump numpy`` output of ``swift_amd.generate``.
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


# --------------------------------------------------------------------------------- store-to-store evaluation (eval/metrics.py:157-280)
# (the reference's schema: which store variables carry a level axis, and the level values its metric names use)
PRESSURE_LEVEL_VARS = ["geopotential", "u_component_of_wind", "v_component_of_wind", "vertical_velocity", "wind_speed", "temperature",
                       "relative_humidity", "specific_humidity", "vorticity", "potential_vorticity"]
DEFAULT_PRESSURE_LEVELS = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
_COORDS = {"time", "number", "prediction_timedelta", "level", "latitude", "longitude"}
_UNIT_NS = {"nanoseconds": 1, "microseconds": 10**3, "milliseconds": 10**6, "seconds": 10**9, "minutes": 60 * 10**9,
            "hours": 3600 * 10**9, "days": 86400 * 10**9}


def _unit_ns(word: str) -> int:
    w = word.strip().lower()
    w = w if w.endswith("s") else w + "s"
    if w not in _UNIT_NS:
        raise ValueError(f"unknown CF time unit {word!r}")
    return _UNIT_NS[w]


def _cf_times(root: str, name: str = "time") -> np.ndarray:
    """A CF-encoded time coordinate (``units: '<unit> since <date>'``, what xarray writes and ``xr.open_zarr`` decodes) -> datetime64[ns]."""
    from ..utils import zarrlite
    v = zarrlite.read_array(root, name)
    if v.dtype.kind == "M":
        return v.astype("datetime64[ns]")
    units = str(zarrlite.read_attrs(root, name).get("units", "nanoseconds since 1970-01-01"))
    unit, _, ref = units.partition(" since ")
    ref_ns = np.datetime64(ref.strip().replace(" ", "T").rstrip("Z") or "1970-01-01", "ns").astype(np.int64)
    ns = np.round(v.astype(np.float64) * _unit_ns(unit)).astype(np.int64) if v.dtype.kind == "f" else v.astype(np.int64) * _unit_ns(unit)
    return (ns + ref_ns).astype("datetime64[ns]")


def _cf_lead_hours(root: str, name: str = "prediction_timedelta") -> list:
    from ..utils import zarrlite
    v = zarrlite.read_array(root, name)
    if v.dtype.kind == "m":
        return [int(x) for x in v.astype("timedelta64[h]").astype(np.int64)]
    unit = str(zarrlite.read_attrs(root, name).get("units", "nanoseconds"))
    return [int(x) for x in (v.astype(np.float64) * _unit_ns(unit) / _UNIT_NS["hours"]).round().astype(np.int64)]


def evaluate_stores(truth: str, pred: str, device=None, sums_fn=None, log=print) -> Dict[str, float]:
    """The reference's ``python -m swift.eval.metrics --truth T.zarr --pred P.zarr`` (eval/metrics.py:157-222): for every lead time of
    the forecast store and every variable (pressure-level variables level by level), latitude-weighted ensemble-mean RMSE, CRPS
    and spread/skill ratio against the truth store's fields at ``init time + lead`` -- keys ``<metric>_<var>[_<level>]_<lead>h``.

    The reference loads both stores whole (``ds[v].values``: 100+ GB for the 15-day 12 x 128 job) and evaluates lead by lead; here
    the forecast store is walked one initial condition at a time (``zarrlite.read_region``: one unit = one chunk file per variable
    in the layout ``generate`` writes), the [lead, member, level, H, W] block of that IC goes to the device once and ONE
    ``swiftk_ensemble_sums`` launch per variable and IC yields the sums of every lead and level; the per-IC sums are combined at
    the end exactly as the reference's batch means are.  ``sums_fn(pred[T, N, V, H, W], y[T, V, H, W], lat) -> [T, V, 4]``
    replaces the device kernel in the host-logic tests."""
    from ..utils import zarrlite
    if sums_fn is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        sums_fn = lambda p, y, lat: ensemble_sums(torch.from_numpy(p).to(dev), torch.from_numpy(y).to(dev), lat).double().cpu().numpy()
    lat = zarrlite.read_array(truth, "latitude").astype(np.float64)
    init_times, truth_times = _cf_times(pred), _cf_times(truth)
    where = {int(t): i for i, t in enumerate(truth_times.astype(np.int64))}
    try:
        init_idx = np.array([where[int(t)] for t in init_times.astype(np.int64)])
    except KeyError as e:
        raise ValueError(f"initial time {np.datetime64(int(e.args[0]), 'ns')} of the forecast store is not in the truth store") from None
    dt_truth = int((truth_times[1] - truth_times[0]).astype("timedelta64[h]").astype(np.int64))
    leads = _cf_lead_hours(pred)
    offs = [h // dt_truth for h in leads]
    if init_idx.max() + max(offs) >= len(truth_times):
        raise ValueError("the truth store ends before the forecasts' last verification time")
    out: Dict[str, float] = {}
    for var in [v for v in zarrlite.list_arrays(pred) if v not in _COORDS]:
        shape, _dt, _dims = zarrlite.array_info(pred, var)
        levelled = var in PRESSURE_LEVEL_VARS
        if levelled != (len(shape) == 6):
            raise ValueError(f"{var}: {len(shape)}-d array, but the reference's schema says it has {'a' if levelled else 'no'} level axis")
        B, N, T = shape[:3]
        L = shape[3] if levelled else 1
        names = [f"{var}_{DEFAULT_PRESSURE_LEVELS[i]}" for i in range(L)] if levelled else [var]  # (by position, as the reference)
        sums = np.zeros((B, T, L, 4))
        for b in range(B):
            p = np.asarray(zarrlite.read_region(pred, var, (b,)), dtype=np.float32)                 # [N, T, (L,) H, W]
            y = np.stack([np.asarray(zarrlite.read_region(truth, var, (int(init_idx[b] + o),)), dtype=np.float32) for o in offs], 0)
            if not levelled:
                p, y = p[:, :, None], y[:, None]
            sums[b] = sums_fn(np.ascontiguousarray(p.transpose(1, 0, 2, 3, 4)), np.ascontiguousarray(y), lat)
        hw = float(shape[-1] * shape[-2])
        for j, h in enumerate(leads):   # the batch statistics of eval/metrics.py:39-134 from the per-IC sums (as generate --metrics)
            s = sums[:, j]
            rmse = np.sqrt(s[..., 0] / hw).mean(0)
            crps = s[..., 1].sum(0) / (B * N * hw) - (s[..., 2] / hw / (2 * N * (N - 1))).mean(0)
            with np.errstate(divide="ignore", invalid="ignore"):
                ssr = np.sqrt(s[..., 3] / hw).mean(0) / rmse
            for i, nm in enumerate(names):
                out[f"crps_{nm}_{h}h"], out[f"rmse_{nm}_{h}h"], out[f"ssr_{nm}_{h}h"] = float(crps[i]), float(rmse[i]), float(ssr[i])
        log(f"{var}: {B} initial conditions x {N} members x {T} lead times" + (f" x {L} levels" if levelled else ""))
    return out


def structure(flat: Dict[str, float]) -> Dict[str, dict]:
    """metric type -> lead time (hours, as a string) -> variable name, as the reference's evaluation_metrics.json (eval/metrics.py:234-252)."""
    res: Dict[str, dict] = {}
    for key, value in flat.items():
        parts = key.split("_")
        res.setdefault(parts[0], {}).setdefault(parts[-1][:-1], {})["_".join(parts[1:-1])] = float(value)
    return res


def main(argv=None):
    """``--truth T.zarr --pred P.zarr``: the reference's CLI and output file (structured evaluation_metrics.json beside the forecast
    store).  ``--pred output.npy --truth truth.npy [--lat lat.npy]``: the ``--dump numpy`` form of ``swift_amd.generate`` (additive)."""
    import time
    ap = argparse.ArgumentParser()
    ap.add_argument("--truth", required=True, help="Path to ground-truth (zarr store; or npy (samples, steps+1, C, H, W) in physical units)")
    ap.add_argument("--pred", required=True, help="Path to prediction (zarr store of swift.generate; or its output-*.npy)")
    ap.add_argument("--lat", default=None, help="npy mode: latitudes (default: linspace(-90, 90, H))")
    ap.add_argument("--interval", type=int, default=6, help="npy mode: hours per step")
    a = ap.parse_args(argv)
    if not a.pred.rstrip("/").endswith(".npy"):
        t0 = time.time()
        flat = evaluate_stores(a.truth, a.pred)
        path = os.path.join(os.path.dirname(os.path.abspath(a.pred.rstrip("/"))), "evaluation_metrics.json")
        with open(path, "w") as f:
            json.dump({"metadata": {"prediction_file": os.path.abspath(a.pred), "truth_file": os.path.abspath(a.truth),
                                    "time": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()), "timestamp": time.time()},
                       "metrics": structure(flat)}, f, indent=2)
        print(f"wrote {path} ({len(flat)} metrics, {time.time() - t0:.1f} s)")
        return flat
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
