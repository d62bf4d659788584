"""A tiny on-disk ERA5 tree in the reference's layout (data/era5.py:58-100: ``<root>/<split>/*.h5`` with ``input/<var>`` 2-D
float32 arrays and an ``input/time`` byte string, ``normalize_{mean,std}.npz``, ``normalize_diff_std_{6,12,24}.npz``,
``lat.npy`` / ``lon.npy``) plus a stand-in ``h5py`` module, because this image has no h5py: the ``*.h5`` files are npz
archives underneath and the stand-in exposes exactly the mapping interface both loaders use (``f["input"][v][()]``,
``f.items()`` / ``group.items()``).  Shared by ``tools/make_golden.py`` (which drives the REFERENCE's ERA5Dataset over the
tree) and ``tests/test_host_logic.py`` (which drives ours over an identical tree)."""
import os
import sys
import types

import numpy as np

VARS = ["2m_temperature", "geopotential_500", "geopotential_850", "temperature_850"]
FORC = ["toa_incident_solar_radiation", "land_sea_mask"]
SHAPE = (6, 12)
N_FILES = 14


class _DS:
    def __init__(self, arr):
        self._a = arr

    def __getitem__(self, key):
        assert key == ()
        a = self._a
        return a[()] if a.ndim == 0 else np.array(a)


class _Group(dict):
    pass


class _File(_Group):
    def __init__(self, path, mode="r"):
        assert mode == "r"
        with np.load(path, allow_pickle=False) as z:
            self["input"] = _Group({k: _DS(z[k]) for k in z.files})

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def install_fake_h5py():
    m = types.ModuleType("h5py")
    m.File = _File
    sys.modules["h5py"] = m
    return m


def _field(seed, name, i, shape):
    rng = np.random.default_rng(seed * 1_000_003 + i * 7919 + sum(map(ord, name)))
    return rng.standard_normal(shape).astype(np.float32)


def write_tree(root, seed=17, split="train"):
    os.makedirs(os.path.join(root, split), exist_ok=True)
    for i in range(N_FILES):
        d = {v: 270.0 + 5.0 * _field(seed, v, i, SHAPE) for v in VARS}
        d.update({f: _field(seed, f, i, SHAPE) for f in FORC})
        if i == 3:
            d[VARS[1]][2, 5] = np.nan  # data/era5.py:60-63 fills NaNs with the field minimum
        t = np.datetime64("2020-01-01T00:00:00") + np.timedelta64(6 * i, "h")
        d["time"] = np.array(str(t).encode("utf-8"))
        with open(os.path.join(root, split, f"{str(t).replace(':', '').replace('-', '')}.h5"), "wb") as f:
            np.savez(f, **d)
    rng = np.random.default_rng(seed)
    np.savez(os.path.join(root, "normalize_mean.npz"), **{k: np.float32(270.0 * (k in VARS) + rng.standard_normal()) for k in VARS + FORC})
    np.savez(os.path.join(root, "normalize_std.npz"), **{k: np.float32(1.0 + abs(rng.standard_normal())) for k in VARS + FORC})
    for dlt in (6, 12, 24):
        np.savez(os.path.join(root, f"normalize_diff_std_{dlt}.npz"), **{k: np.float32(0.5 + abs(rng.standard_normal())) for k in VARS})
    np.save(os.path.join(root, "lat.npy"), np.linspace(-90, 90, SHAPE[0]))
    np.save(os.path.join(root, "lon.npy"), np.linspace(0, 360, SHAPE[1], endpoint=False))
    return root
