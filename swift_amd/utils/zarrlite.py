"""Zarr v2 directory stores written (and read back) with the standard library + numpy.

The reference's default output of ``swift.generate`` is a zarr group created by
``utils/io.py:161-235`` (``fast_create_empty_zarr``) and filled per (sample, member) in
``generate.py:141-152``; this image has neither ``zarr`` nor ``xarray``, and the format is
simple enough not to need them: a store is a directory, a group is ``.zgroup`` + ``.zattrs``
JSON, an array is ``<name>/.zarray`` + ``.zattrs`` JSON plus one file per chunk named by the
dot-joined chunk index holding the chunk's C-order bytes (``compressor: null``).  xarray's
conventions are kept (``_ARRAY_DIMENSIONS`` attributes, CF-encoded time coordinates,
consolidated ``.zmetadata``), so ``xr.open_zarr(path, decode_timedelta=True)`` and
``zarr.open_group(path)`` read these stores where those packages exist.

Layout written by :func:`create_forecast_store` (identical to the reference's):
one array per variable, dims ``(time, number, prediction_timedelta, [level,] latitude, longitude)``
float32, chunks ``(batch, 1, steps + 1, [levels,] lat, lon)`` -- with ``batch = 1`` the whole
trajectory of one (initial condition, member) unit is exactly one chunk file per variable,
so ranks write disjoint files and never read-modify-write.
"""
from __future__ import annotations

import json
import os
import re
from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def compress_variables(variables: Sequence[str]) -> Dict[str, List[int]]:
    """``name_<level>`` channels grouped per variable, in order of appearance (utils/io.py:73-82)."""
    out: Dict[str, List[int]] = defaultdict(list)
    for var in variables:
        m = re.match(r"^(.*)_(\d+)$", var)
        if m:
            out[m.group(1)].append(int(m.group(2)))
        else:
            out[var] = []
    return dict(out)


def variable_channels(variables: Sequence[str]) -> Dict[str, List[int]]:
    """Channel indices of each store variable (generate.py:62-72)."""
    res, k = {}, 0
    for var, levels in compress_variables(variables).items():
        n = max(len(levels), 1)
        res[var] = list(range(k, k + n))
        k += n
    return res


def _dump(path: str, obj) -> None:
    with open(path, "w") as f:
        json.dump(obj, f, indent=1)


def _dtype_str(dt: np.dtype) -> str:
    dt = np.dtype(dt)
    return dt.str if dt.byteorder != "|" else "|" + dt.str[1:]


def create_group(path: str, attrs: Optional[dict] = None) -> None:
    os.makedirs(path, exist_ok=True)
    _dump(os.path.join(path, ".zgroup"), {"zarr_format": 2})
    _dump(os.path.join(path, ".zattrs"), attrs or {})


def create_array(root: str, name: str, shape: Sequence[int], chunks: Sequence[int], dtype, dims: Sequence[str],
                 fill_value=0.0, attrs: Optional[dict] = None) -> None:
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    _dump(os.path.join(d, ".zarray"), {"zarr_format": 2, "shape": [int(s) for s in shape], "chunks": [int(c) for c in chunks],
                                        "dtype": _dtype_str(dtype), "compressor": None, "fill_value": fill_value,
                                        "order": "C", "filters": None})
    _dump(os.path.join(d, ".zattrs"), {"_ARRAY_DIMENSIONS": list(dims), **(attrs or {})})


_META_CACHE: Dict[Tuple[str, str], Tuple[tuple, np.dtype]] = {}


def _chunk_meta(root: str, name: str):
    key = (root, name)
    if key not in _META_CACHE:
        with open(os.path.join(root, name, ".zarray")) as f:
            meta = json.load(f)
        _META_CACHE[key] = (tuple(meta["chunks"]), np.dtype(meta["dtype"]))
    return _META_CACHE[key]


def write_chunk(root: str, name: str, index: Sequence[int], data: np.ndarray) -> None:
    """Write one WHOLE chunk (``data.shape`` == the array's chunk shape; edge chunks are padded by the caller)."""
    chunks, dt = _chunk_meta(root, name)
    assert tuple(data.shape) == chunks, (data.shape, chunks)
    data = np.ascontiguousarray(data, dtype=dt)
    tmp = os.path.join(root, name, "." + ".".join(str(int(i)) for i in index) + f".{os.getpid()}.tmp")
    with open(tmp, "wb") as f:
        f.write(memoryview(data).cast("B"))
    os.replace(tmp, os.path.join(root, name, ".".join(str(int(i)) for i in index)))


def write_full(root: str, name: str, data: np.ndarray, dims: Sequence[str], attrs: Optional[dict] = None) -> None:
    """A small array (coordinates) as a single chunk."""
    data = np.asarray(data)
    fill = "NaN" if data.dtype.kind == "f" else 0
    create_array(root, name, data.shape, data.shape if data.size else (1,) * data.ndim, data.dtype, dims, fill_value=fill, attrs=attrs)
    if data.size:
        write_chunk(root, name, (0,) * data.ndim, data)


def consolidate(root: str) -> None:
    """``.zmetadata`` (zarr.convenience.consolidate_metadata, generate.py:281-285)."""
    meta = {}
    for dirpath, _dirs, files in os.walk(root):
        for fn in files:
            if fn in (".zgroup", ".zarray", ".zattrs"):
                key = os.path.relpath(os.path.join(dirpath, fn), root).replace(os.sep, "/")
                with open(os.path.join(dirpath, fn)) as f:
                    meta[key] = json.load(f)
    _dump(os.path.join(root, ".zmetadata"), {"zarr_consolidated_format": 1, "metadata": dict(sorted(meta.items()))})


def create_forecast_store(ofile: str, variables: Sequence[str], times: np.ndarray, lat: np.ndarray, lon: np.ndarray, members: int,
                          steps: int, interval: int = 6, batch: int = 1) -> Dict[str, List[int]]:
    """Empty forecast store with the reference's structure (utils/io.py:161-235); returns {variable: channel indices}."""
    create_group(ofile)
    n = len(times)
    t_ns = np.asarray(times, dtype="datetime64[ns]").astype(np.int64)
    write_full(ofile, "time", t_ns, ["time"], {"units": "nanoseconds since 1970-01-01", "calendar": "proleptic_gregorian"})
    td = (np.arange(steps + 1, dtype=np.int64) * interval * 3_600_000_000_000)
    write_full(ofile, "prediction_timedelta", td, ["prediction_timedelta"], {"units": "nanoseconds", "dtype": "timedelta64[ns]"})
    write_full(ofile, "latitude", np.asarray(lat, np.float32), ["latitude"])
    write_full(ofile, "longitude", np.asarray(lon, np.float32), ["longitude"])
    write_full(ofile, "number", np.arange(members, dtype=np.int64), ["number"])
    comp = compress_variables(variables)
    if any(len(lv) for lv in comp.values()):
        write_full(ofile, "level", np.arange(max(len(lv) for lv in comp.values()), dtype=np.int64), ["level"])
    n_lat, n_lon = len(lat), len(lon)
    for var, levels in comp.items():
        if levels:
            shape, chunks = (n, members, steps + 1, len(levels), n_lat, n_lon), (batch, 1, steps + 1, len(levels), n_lat, n_lon)
            dims = ["time", "number", "prediction_timedelta", "level", "latitude", "longitude"]
        else:
            shape, chunks = (n, members, steps + 1, n_lat, n_lon), (batch, 1, steps + 1, n_lat, n_lon)
            dims = ["time", "number", "prediction_timedelta", "latitude", "longitude"]
        create_array(ofile, var, shape, chunks, np.float32, dims, fill_value=0.0)
    return variable_channels(variables)


def write_unit(ofile: str, var_channels: Dict[str, List[int]], sample: int, member: int, traj: np.ndarray) -> None:
    """One (sample, member) trajectory ``traj [steps+1, C, H, W]`` -> one chunk per variable (chunk batch 1), generate.py:143-152."""
    for var, ch in var_channels.items():
        if len(ch) == 1 and not _has_level(ofile, var):
            write_chunk(ofile, var, (sample, member, 0, 0, 0), traj[None, None, :, ch[0]])
        else:
            write_chunk(ofile, var, (sample, member, 0, 0, 0, 0), traj[None, None][:, :, :, ch])


def write_unit_step(ofile: str, var_channels: Dict[str, List[int]], sample: int, member: int, step: int, fields: np.ndarray) -> None:
    """Lead step ``step`` of one (sample, member) unit, ``fields [C, H, W]``: written in place into the unit's chunk files
    (chunk batch 1, uncompressed: a chunk file is the raw ``[steps+1, (levels,) H, W]`` block, so a step is one contiguous
    range of it).  A chunk becomes complete when its last step has been written; ``write_unit`` is the all-at-once form."""
    for var, ch in var_channels.items():
        chunks, dt = _chunk_meta(ofile, var)
        data = np.ascontiguousarray(fields[ch[0]] if len(chunks) == 5 else fields[ch], dtype=dt)
        name = ".".join(str(int(i)) for i in ((sample, member) + (0,) * (len(chunks) - 2)))
        fd = os.open(os.path.join(ofile, var, name), os.O_WRONLY | os.O_CREAT, 0o644)
        try:
            os.pwrite(fd, memoryview(data).cast("B"), int(step) * data.nbytes)
        finally:
            os.close(fd)


def _has_level(ofile: str, var: str) -> bool:
    return len(_chunk_meta(ofile, var)[0]) == 6


# ------------------------------------------------------------------------------------------ reader (tests, eval)
def _blosc1_decode(buf: bytes) -> bytes:
    """One Blosc-1 chunk -> its bytes, without the c-blosc library: the container is parsed here, the inner streams are decoded by
    real codec implementations (LZ4 raw blocks, Zstandard and Snappy through ``pyarrow``, zlib through the standard library).

    Layout, restated from c-blosc 1.x (blosc.h / blosc.c; numcodecs' ``Blosc`` -- zarr's default compressor, which wrote the
    reference's stores, utils/io.py:161-235 -- emits exactly this): a 16-byte header [version = 2, versionlz, flags, typesize,
    nbytes i32, blocksize i32, cbytes i32]; flags bit 0 = byte shuffle, bit 1 = stored uncompressed (payload follows the header),
    bit 2 = bit shuffle, bit 4 = blocks are not split, bits 5-7 = codec (0 BloscLZ, 1 LZ4 / LZ4HC, 2 Snappy, 3 zlib, 4 Zstd); then one
    i32 start offset per block; a block is ``typesize`` streams (one per byte position of the shuffled elements) unless bit 4 is
    set, it is the short last block, typesize > 16 or a stream would be under 128 bytes -- then one stream; a stream is an i32
    length followed by its bytes, stored raw when the length equals the stream's uncompressed size.  NOT validated against c-blosc
    in this image (neither it nor numcodecs is installed): the tests assemble chunks by this description around real LZ4 / Zstd
    streams.  Every size is cross-checked, so a chunk that does not follow the description raises instead of decoding to garbage."""
    import struct
    if len(buf) < 16:
        raise ValueError("blosc: chunk shorter than its header")
    version, _versionlz, flags, typesize = buf[0], buf[1], buf[2], buf[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<iii", buf, 4)
    if version != 2:
        raise NotImplementedError(f"blosc: chunk format version {version} (this reader knows the Blosc-1 format, version 2)")
    if nbytes < 0 or blocksize <= 0 and nbytes > 0 or cbytes > len(buf) or typesize < 1:
        raise ValueError("blosc: inconsistent chunk header")
    if nbytes == 0:
        return b""
    if flags & 0x2:
        if 16 + nbytes > len(buf):
            raise ValueError("blosc: stored chunk shorter than its header says")
        return bytes(buf[16:16 + nbytes])
    if flags & 0x4:
        raise NotImplementedError("blosc: bit-shuffled chunks are not supported (byte shuffle and no shuffle are)")
    fmt = (flags & 0xE0) >> 5
    names = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}
    if fmt == 0 or fmt not in names:
        raise NotImplementedError(f"blosc: inner codec {names.get(fmt, fmt)!r} is not supported without the c-blosc library (lz4, lz4hc, zstd, "
                                  "zlib and snappy are)")

    def inner(data: bytes, size: int) -> bytes:
        if fmt == 3:
            import zlib
            out = zlib.decompress(data)
        else:
            import pyarrow as pa
            out = pa.decompress(data, decompressed_size=size, codec={1: "lz4_raw", 2: "snappy", 4: "zstd"}[fmt], asbytes=True)
        if len(out) != size:
            raise ValueError("blosc: a stream decoded to the wrong size")
        return out

    nblocks = (nbytes + blocksize - 1) // blocksize
    if 16 + 4 * nblocks > len(buf):
        raise ValueError("blosc: block offsets run past the chunk")
    bstarts = struct.unpack_from(f"<{nblocks}i", buf, 16)
    dont_split = bool(flags & 0x10)
    out = bytearray(nbytes)
    for i in range(nblocks):
        bsize = min(blocksize, nbytes - i * blocksize)
        split = (not dont_split) and bsize == blocksize and typesize <= 16 and blocksize // typesize >= 128
        nsplits = typesize if split else 1
        neblock = bsize // nsplits
        pos, parts = bstarts[i], []
        for _ in range(nsplits):
            if pos < 16 + 4 * nblocks or pos + 4 > len(buf):
                raise ValueError("blosc: stream offset outside the chunk")
            (cb,) = struct.unpack_from("<i", buf, pos)
            pos += 4
            if cb < 0 or pos + cb > len(buf):
                raise ValueError("blosc: stream length outside the chunk")
            parts.append(bytes(buf[pos:pos + cb]) if cb == neblock else inner(bytes(buf[pos:pos + cb]), neblock))
            pos += cb
        block = b"".join(parts)
        if len(block) != bsize:
            raise ValueError("blosc: a block decoded to the wrong size")
        if (flags & 0x1) and typesize > 1:
            n = bsize // typesize
            body = np.frombuffer(block, dtype=np.uint8, count=n * typesize).reshape(typesize, n).T.tobytes()
            block = body + block[n * typesize:]
        out[i * blocksize:i * blocksize + bsize] = block
    return bytes(out)


def _chunk_decoder(meta: dict, where: str):
    """bytes -> bytes for one chunk of an array with this ``.zarray``: identity for the stores this package writes
    (``compressor: null``), the standard library for the zlib / gzip / bz2 / lzma codec ids, ``numcodecs`` -- when it is
    importable -- for everything else a zarr v2 store may carry.  The reference's stores are written by xarray / zarr with the
    default ``Blosc(cname="lz4", clevel=5, shuffle=1)`` (utils/io.py:161-235): without numcodecs those chunks go through
    ``_blosc1_decode`` (container parsed here, inner streams by pyarrow's codecs); anything else is refused with an error that names
    the compressor and the two ways out instead of handing back garbage."""
    comp, filters = meta.get("compressor"), meta.get("filters") or []
    if comp is None and not filters:
        return lambda b: b
    stdlib = {"zlib": ("zlib", "decompress"), "gzip": ("gzip", "decompress"), "bz2": ("bz2", "decompress"), "lzma": ("lzma", "decompress")}
    if not filters and comp.get("id") in stdlib and not (comp.get("id") == "lzma" and (comp.get("format", 1) != 1 or comp.get("filters"))):
        import importlib
        mod, fn = stdlib[comp["id"]]
        return getattr(importlib.import_module(mod), fn)
    try:
        import numcodecs
    except ImportError:
        if not filters and comp is not None and comp.get("id") == "blosc":
            try:
                import pyarrow  # noqa: F401  (the inner LZ4 / Zstd / Snappy streams)
                return _blosc1_decode
            except ImportError:
                pass
        desc = ", ".join(f"{k}={v!r}" for k, v in (comp or {}).items()) or "none"
        fl = "; filters: " + ", ".join(str(f.get("id")) for f in filters) if filters else ""
        raise NotImplementedError(
            f"{where}: chunks are compressed with [{desc}]{fl}, which needs the `numcodecs` package (not installed here).  Install "
            "numcodecs, or rewrite the store uncompressed (zarr: `compressor=None`; xarray: `encoding={var: {'compressor': None}}`) -- "
            "swift_amd.generate writes `compressor: null` stores, which any zarr reader opens") from None
    codecs = [numcodecs.get_codec(f) for f in filters]
    cz = numcodecs.get_codec(comp) if comp is not None else None

    def decode(b):
        if cz is not None:
            b = cz.decode(b)
        for f in reversed(codecs):
            b = f.decode(b)
        return b

    return decode


def read_array(root: str, name: str) -> np.ndarray:
    """Whole array from a v2 store (missing chunks = fill_value); compressed chunks as ``_chunk_decoder`` allows."""
    with open(os.path.join(root, name, ".zarray")) as f:
        meta = json.load(f)
    decode = _chunk_decoder(meta, os.path.join(root, name))
    raw = meta.get("compressor") is None and not meta.get("filters")
    if meta.get("order", "C") != "C":
        raise NotImplementedError(f"{os.path.join(root, name)}: Fortran-ordered chunks are not supported")
    shape, chunks, dt = tuple(meta["shape"]), tuple(meta["chunks"]), np.dtype(meta["dtype"])
    fill = meta.get("fill_value")
    fill = np.nan if fill == "NaN" else (0 if fill is None else fill)
    out = np.full(shape, fill, dtype=dt)
    grid = [range((s + c - 1) // c) for s, c in zip(shape, chunks)] if shape else []
    sep = meta.get("dimension_separator", ".")
    for idx in np.ndindex(*[len(g) for g in grid]):
        p = os.path.join(root, name, sep.join(str(i) for i in idx))
        if not os.path.exists(p):
            continue
        if raw:
            blk = np.fromfile(p, dtype=dt).reshape(chunks)
        else:
            with open(p, "rb") as f:
                blk = np.frombuffer(bytes(decode(f.read())), dtype=dt).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out


def read_attrs(root: str, name: str = "") -> dict:
    with open(os.path.join(root, name, ".zattrs")) as f:
        return json.load(f)


def list_arrays(root: str) -> List[str]:
    """Names of the arrays of a (flat) group: sub-directories with a ``.zarray`` (consolidated metadata is not needed)."""
    return sorted(n for n in os.listdir(root) if os.path.isfile(os.path.join(root, n, ".zarray")))


def array_info(root: str, name: str) -> Tuple[tuple, np.dtype, List[str]]:
    """(shape, dtype, dimension names) of an array; dimension names from xarray's ``_ARRAY_DIMENSIONS`` attribute ([] without)."""
    with open(os.path.join(root, name, ".zarray")) as f:
        meta = json.load(f)
    try:
        dims = list(read_attrs(root, name).get("_ARRAY_DIMENSIONS", []))
    except OSError:
        dims = []
    return tuple(meta["shape"]), np.dtype(meta["dtype"]), dims


def read_region(root: str, name: str, sel: Sequence) -> np.ndarray:
    """``array[sel]`` for a tuple of ints / slices (step 1) over the leading axes, touching only the chunks the selection meets --
    what an evaluation that walks a 100-GB forecast store one initial condition at a time needs (``read_array`` loads everything).
    Integer entries drop their axis, as in numpy."""
    with open(os.path.join(root, name, ".zarray")) as f:
        meta = json.load(f)
    decode = _chunk_decoder(meta, os.path.join(root, name))
    raw = meta.get("compressor") is None and not meta.get("filters")
    if meta.get("order", "C") != "C":
        raise NotImplementedError(f"{os.path.join(root, name)}: Fortran-ordered chunks are not supported")
    shape, chunks, dt = tuple(meta["shape"]), tuple(meta["chunks"]), np.dtype(meta["dtype"])
    fill = meta.get("fill_value")
    fill = np.nan if fill == "NaN" else (0 if fill is None else fill)
    sel = tuple(sel) + (slice(None),) * (len(shape) - len(sel))
    lo, hi, drop = [], [], []
    for ax, (s_, n) in enumerate(zip(sel, shape)):
        if isinstance(s_, (int, np.integer)):
            i = int(s_) + (n if s_ < 0 else 0)
            if not 0 <= i < n:
                raise IndexError(f"{name}: index {s_} out of range for axis {ax} of size {n}")
            lo.append(i); hi.append(i + 1); drop.append(ax)
        else:
            a, b, st = s_.indices(n)
            if st != 1:
                raise NotImplementedError("read_region: slices with a step")
            lo.append(a); hi.append(max(a, b))
    out = np.full([h - l for l, h in zip(lo, hi)], fill, dtype=dt)
    sep = meta.get("dimension_separator", ".")
    grid = [range(l // c, (h - 1) // c + 1) if h > l else range(0) for l, h, c in zip(lo, hi, chunks)]
    for idx in np.ndindex(*[len(g) for g in grid]):
        cidx = [g[i] for g, i in zip(grid, idx)]
        p = os.path.join(root, name, sep.join(str(i) for i in cidx))
        if not os.path.exists(p):
            continue
        if raw:
            blk = np.memmap(p, dtype=dt, mode="r", shape=chunks) if os.path.getsize(p) == int(np.prod(chunks)) * dt.itemsize else \
                np.fromfile(p, dtype=dt).reshape(chunks)
        else:
            with open(p, "rb") as f:
                blk = np.frombuffer(bytes(decode(f.read())), dtype=dt).reshape(chunks)
        src, dst = [], []
        for ci, c, l, h in zip(cidx, chunks, lo, hi):
            a, b = max(l, ci * c), min(h, (ci + 1) * c)
            src.append(slice(a - ci * c, b - ci * c))
            dst.append(slice(a - l, b - l))
        out[tuple(dst)] = blk[tuple(src)]
    return out.reshape([n for ax, n in enumerate(out.shape) if ax not in drop]) if drop else out
