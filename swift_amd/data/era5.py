"""ERA5 dataset interface of the forecast path (mirrors reference src/swift/data/era5.py:11-227).

The rollout and the losses only touch a dataset through
``standardize_x / unstandardize_x / standardize_t / unstandardize_t / zero_field / get_forcings``
and a few attributes (``variables, forcings, intervals, residual, n_target_channels,
n_condition_channels, img_resolution, _shape``).  ``ERA5Dataset`` keeps that surface and the
reference's on-disk layout (one h5 per time step with ``input/<var>`` 2-D float32 arrays,
``normalize_{mean,std}.npz``, ``normalize_diff_std_{6,12,24}.npz``; era5.py:58-100);
``SyntheticERA5Dataset`` is the same interface over seeded N(0,1) fields for machines without
ERA5 (the benchmark and the tests) -- select it with
``data.dataset._target_=swift_amd.data.era5.SyntheticERA5Dataset``.
"""
from __future__ import annotations

import os
from glob import glob
from typing import Dict, Sequence, Tuple, Union

import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils.detinit import det_normal


class _ERA5Base(Dataset):
    variables: Sequence[str]
    forcings: Sequence[str]
    intervals: Sequence[int]
    residual: bool
    x_means: np.ndarray
    x_stds: np.ndarray
    t_means: Union[Dict[int, np.ndarray], np.ndarray]
    t_stds: Union[Dict[int, np.ndarray], np.ndarray]
    _shape: Tuple[int, int, int]

    @property
    def n_target_channels(self) -> int:
        return self._shape[0]

    @property
    def n_condition_channels(self) -> int:
        return self.n_target_channels + len(self.forcings)

    @property
    def img_resolution(self) -> Tuple[int, int]:
        return self._shape[1], self._shape[2]

    # -- standardisation (era5.py:110-166) ---------------------------------------------------
    def _transform_standardize(self, v, means, stds, inverse: bool = False):
        if isinstance(v, torch.Tensor):
            m = torch.as_tensor(means, device=v.device, dtype=v.dtype)
            s = torch.as_tensor(stds, device=v.device, dtype=v.dtype)
        else:
            m, s = means, stds
        channels = v.shape[1 if v.ndim == 4 else 0]
        nv, nf = len(self.variables), len(self.forcings)
        if channels == nv:  # stats cover variables + forcings; pick by channel count
            m, s = m[:nv], s[:nv]
        elif channels == nf:
            m, s = m[nv:], s[nv:]
        return v * s + m if inverse else (v - m) / s

    def zero_field(self, x, delta: int = 6):
        channels = x.shape[1 if x.ndim == 4 else 0]
        if delta == 24 or "sea_surface_temperature" not in self.variables or channels == len(self.forcings):
            return x
        idx = list(self.variables).index("sea_surface_temperature")
        if x.ndim == 4:
            x[:, idx, ...] = 0
        elif x.ndim == 3:
            x[idx, ...] = 0
        return x

    def standardize_x(self, x, delta: int = 6):
        return self.zero_field(self._transform_standardize(x, self.x_means, self.x_stds), delta)

    def unstandardize_x(self, x, delta: int = 6):
        return self.zero_field(self._transform_standardize(x, self.x_means, self.x_stds, inverse=True), delta)

    def _t_stats(self, delta):
        if isinstance(self.t_stds, dict):
            return self.t_means[delta], self.t_stds[delta]
        return self.t_means, self.t_stds

    def standardize_t(self, t, delta: int = 6):
        return self.zero_field(self._transform_standardize(t, *self._t_stats(delta)), delta)

    def unstandardize_t(self, t, delta: int = 6):
        return self.zero_field(self._transform_standardize(t, *self._t_stats(delta), inverse=True), delta)

    # -- flat per-channel vectors for the fused rollout kernel -------------------------------
    def rollout_stats(self, delta: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(mean_x, std_x, std_t) over the target channels with ONE delta everywhere, zero_field folded in.

        This is what the multistep CRPS loss applies (loss.py:402-406: ``unstandardize_t``, ``unstandardize_x`` and
        ``standardize_x`` all take the batch's delta): with delta != 24 the SST channel is zeroed by every one of the three
        calls (mean 0 / std 1 / residual-std 0: every quantity of the channel stays 0); with delta == 24 nothing is zeroed and
        the channel keeps its real statistics.  The generate / validation rollout mixes deltas (state at the DEFAULT delta,
        residual at the interval): ``swift_amd.rollout.update_stats`` derives its vectors from these.
        """
        key = (int(delta), str(device))
        cache = self.__dict__.setdefault("_rollout_stats_cache", {})
        if key in cache:  # (three small host-to-device copies per call otherwise, each a stream synchronisation)
            return cache[key]
        nv = len(self.variables)
        mx = torch.as_tensor(np.asarray(self.x_means), dtype=torch.float32).reshape(-1)[:nv].clone()
        sx = torch.as_tensor(np.asarray(self.x_stds), dtype=torch.float32).reshape(-1)[:nv].clone()
        tm, ts = self._t_stats(delta)
        if float(np.abs(np.asarray(tm).reshape(-1)[:nv]).max()) != 0.0 and self.residual:
            raise ValueError("residual targets are expected to have zero mean (era5.py:100)")
        st = torch.as_tensor(np.asarray(ts), dtype=torch.float32).reshape(-1)[:nv].clone()
        if "sea_surface_temperature" in self.variables and delta != 24:
            i = list(self.variables).index("sea_surface_temperature")
            mx[i], sx[i], st[i] = 0.0, 1.0, 0.0
        cache[key] = (mx.to(device), sx.to(device), st.to(device))
        return cache[key]


class ERA5Dataset(_ERA5Base):
    """h5-backed dataset with the reference's layout (needs ``h5py``, which this image lacks)."""

    def __init__(self, root: str, variables, forcings=(), intervals=(6, 12, 24), split: str = "train",
                 residual: bool = False):
        super().__init__()
        assert sorted(intervals) in ([6], [12], [24], [6, 12], [6, 24], [12, 24], [6, 12, 24])
        try:
            import h5py  # noqa: F401
        except ImportError as e:  # pragma: no cover
            raise ImportError("ERA5Dataset reads h5 files and needs h5py; use SyntheticERA5Dataset without it") from e
        self.root, self.variables, self.forcings = root, list(variables), list(forcings)
        self.files = sorted(glob(os.path.join(root, split, "*.h5")))
        self.intervals, self.residual = list(intervals), residual
        ld = lambda fn, vs: np.stack([np.load(os.path.join(root, fn))[v] for v in vs], 0).reshape(-1, 1, 1)
        self.x_means = ld("normalize_mean.npz", self.variables + self.forcings)
        self.x_stds = ld("normalize_std.npz", self.variables + self.forcings)
        if residual:
            self.t_stds = {i: ld(f"normalize_diff_std_{i}.npz", self.variables) for i in self.intervals}
            self.t_means = {i: np.zeros_like(self.t_stds[i]) for i in self.intervals}
        else:
            self.t_means, self.t_stds = self.x_means, self.x_stds
        self._shape = self._load_file(self.files[0], self.variables).shape

    def _load_file(self, path, variables):
        import h5py

        def fill(v):
            if np.isnan(v).any():
                np.copyto(v, np.nanmin(v), where=np.isnan(v))
            return v

        with h5py.File(path, "r") as f:
            return np.stack([fill(f["input"][v][()]) for v in variables], axis=0)

    def get_forcings(self, idx: int) -> torch.Tensor:
        return torch.from_numpy(self._load_file(self.files[idx], self.forcings)).float()

    def get_state(self, idx: int) -> torch.Tensor:
        """Physical fields of time step ``idx`` [n_vars, H, W]: one file read (``self[idx]`` reads two: x and its target)."""
        return torch.from_numpy(self._load_file(self.files[idx], self.variables)).float()

    def get_lat_lon(self):
        """data/era5.py:172-175."""
        return (np.load(os.path.join(self.root, "lat.npy")).astype(np.float32),
                np.load(os.path.join(self.root, "lon.npy")).astype(np.float32))

    def get_time(self, idx: int) -> np.datetime64:
        """data/era5.py:177-181."""
        import h5py
        with h5py.File(self.files[idx], "r") as f:
            timestamp = f["input"]["time"][()]
            assert isinstance(timestamp, bytes)
            return np.datetime64(timestamp.decode("utf-8"))

    def __len__(self):
        return len(self.files[: -(max(self.intervals) * 1 // 6)])

    def __getitem__(self, spec):
        if isinstance(spec, tuple):
            spec = tuple(int(i) for i in spec)
            idx, offset = spec[0], spec[1]
            delta = spec[2] if len(spec) > 2 else None
        else:
            idx, offset, delta = int(spec), 1, None
        if delta is None:
            delta = np.random.choice(self.intervals)
        x = self._load_file(self.files[idx], self.variables + self.forcings)
        t = self._load_file(self.files[idx + (offset * delta // 6)], self.variables)
        if self.residual:
            nv = len(self.variables)
            prev = self._load_file(self.files[idx + (offset - 1) * delta // 6], self.variables) if offset > 1 else x[:nv]
            t = t - prev
        x = torch.from_numpy(self.standardize_x(x, delta)).float()
        t = torch.from_numpy(self.standardize_t(t, delta)).float()
        return (x, t), (idx, torch.tensor(delta / 10.0).float())


class SyntheticERA5Dataset(_ERA5Base):
    """Seeded synthetic fields with the ERA5Dataset interface.

    State / forcing "files" are N(0,1) draws keyed by the time index (generated on demand, so
    the dataset has no memory footprint); statistics default to mean 0 / std 1 for x and a
    residual std of 0.1 so that long rollouts stay bounded.
    """

    def __init__(self, variables, forcings=(), img_resolution=(128, 256), length: int = 1024, intervals=(6, 12, 24),
                 split: str = "train", residual: bool = True, seed: int = 1234, root: str = "", t_std: float = 0.1,
                 random_stats: bool = False):
        super().__init__()
        self.variables, self.forcings = list(variables), list(forcings)
        self.intervals, self.residual, self.seed, self.length = list(intervals), residual, seed, length
        nv, nf = len(self.variables), len(self.forcings)
        self._shape = (nv, int(img_resolution[0]), int(img_resolution[1]))
        if random_stats:
            self.x_means = det_normal((nv + nf, 1, 1), seed, "x_mean", std=2.0).numpy()
            self.x_stds = (det_normal((nv + nf, 1, 1), seed, "x_std", std=0.3).abs() + 0.5).numpy()
            self.t_stds = {i: (det_normal((nv, 1, 1), seed, f"t_std{i}", std=0.05).abs() + t_std).numpy()
                           for i in self.intervals}
        else:
            self.x_means = np.zeros((nv + nf, 1, 1), np.float32)
            self.x_stds = np.ones((nv + nf, 1, 1), np.float32)
            self.t_stds = {i: np.full((nv, 1, 1), t_std, np.float32) for i in self.intervals}
        self.t_means = {i: np.zeros_like(self.t_stds[i]) for i in self.intervals}
        if not residual:
            self.t_means, self.t_stds = self.x_means, self.x_stds

    def _fields(self, idx: int, kind: str, n: int) -> torch.Tensor:
        return det_normal((n, self._shape[1], self._shape[2]), self.seed, f"{kind}{int(idx)}")

    def get_lat_lon(self):
        H, W = self._shape[1], self._shape[2]
        return np.linspace(-90.0, 90.0, H, dtype=np.float32), np.linspace(0.0, 360.0, W, endpoint=False, dtype=np.float32)

    def get_time(self, idx: int) -> np.datetime64:
        return np.datetime64("2020-01-01T00:00:00") + np.timedelta64(6 * int(idx), "h")

    def get_forcings(self, idx: int) -> torch.Tensor:
        # a small most-recently-used cache stands in for the OS page cache an h5-backed dataset has: the multistep loss asks
        # for the same (index, lead step) files again and again (loss.py:378-392), and a forcing slab is 393 KB
        cache = self.__dict__.setdefault("_forc_cache", {})
        idx = int(idx)
        f = cache.pop(idx, None)
        if f is None:
            f = self._fields(idx, "forc", len(self.forcings))
            if len(cache) >= 256:
                cache.pop(next(iter(cache)))
        cache[idx] = f  # (re-inserted last: dict order = recency)
        return f.clone()

    def get_state(self, idx: int) -> torch.Tensor:
        return self._fields(idx, "state", len(self.variables))

    def __len__(self):
        return self.length - (max(self.intervals) // 6)

    def __getitem__(self, spec):
        if isinstance(spec, tuple):
            spec = tuple(int(i) for i in spec)
            idx, offset = spec[0], spec[1]
            delta = spec[2] if len(spec) > 2 else None
        else:
            idx, offset, delta = int(spec), 1, None
        if delta is None:
            delta = int(np.random.choice(self.intervals))
        nv = len(self.variables)
        x = torch.cat([self._fields(idx, "state", nv), self.get_forcings(idx)], 0)
        t = self._fields(idx + offset * delta // 6, "state", nv)
        if self.residual:
            prev = self._fields(idx + (offset - 1) * delta // 6, "state", nv) if offset > 1 else x[:nv]
            t = t - prev
        x = self.standardize_x(x.numpy(), delta)
        t = self.standardize_t(t.numpy(), delta)
        return (torch.from_numpy(np.asarray(x)).float(), torch.from_numpy(np.asarray(t)).float()), \
               (idx, torch.tensor(delta / 10.0).float())


class SyntheticERA5RollOutDataset(SyntheticERA5Dataset):
    """Validation items of the reference's ERA5RollOutDataset (data/era5.py:230-256) on the synthetic fields:
    ``(x standardised [C,H,W], targets physical [interval/4 + 1, C, H, W] = (6 h, day 1, day 2, ...), idx)``."""

    def __init__(self, interval: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.interval = int(interval)

    def __len__(self):
        return self.length - self.interval

    def __getitem__(self, idx: int):
        idx = int(idx)
        nv = len(self.variables)
        assert self.interval >= 4, "cannot even predict one day"
        x = torch.from_numpy(np.asarray(self.standardize_x(self._fields(idx, "state", nv).numpy()))).float()
        ts = [self._fields(idx + 1, "state", nv)] + [self._fields(i, "state", nv) for i in range(idx + 4, idx + 4 + self.interval, 4)]
        return x, torch.stack(ts, 0).float(), idx


class ERA5RollOutDataset(ERA5Dataset):
    """data/era5.py:230-256: validation items ``(x standardised, targets physical [interval/4 + 1, C, H, W], idx)``."""

    def __init__(self, interval: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.interval = int(interval)

    def __len__(self):
        return len(self.files[: -self.interval])

    def __getitem__(self, idx: int):
        idx = int(idx)
        assert self.interval >= 4, "cannot even predict one day"
        x = torch.from_numpy(self.standardize_x(self._load_file(self.files[idx], self.variables))).float()
        ts = [self._load_file(self.files[idx + 1], self.variables)]
        ts += [self._load_file(self.files[i], self.variables) for i in range(idx + 4, idx + 4 + self.interval, 4)]
        return x, torch.from_numpy(np.stack(ts, 0)).float(), idx
