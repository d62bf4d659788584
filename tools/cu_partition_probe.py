#!/usr/bin/env python
"""Can the chip be partitioned between the matrix kernels and the HBM-bound ModulatedNorm pass?

Two HIP streams with CU masks (hipExtStreamCreateWithCUMask): a "matrix" stream on 256 - R CUs and a "stream" stream on the R
reserved CUs (R / 8 per XCD).  Measured at `units` units per launch (default 48 = half of the bench's step):
  1. where the masks land (XCC / SE / CU of every workgroup of a probe kernel, tools/cu_probe/where.hip),
  2. w1 + SwiGLU, w2 and the fused to_qkv + attention kernel on all 256 CUs against 256 - R CUs (persistent grid = CUs),
  3. the pair-form ModulatedNorm on R CUs alone,
  4. both at the same time: each stream's own elapsed time and the wall time of the pair.

usage: cu_partition_probe.py [units] [reserved_per_xcd ...]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from swift_amd import _lib, ops  # noqa: E402

UNITS = int(sys.argv[1]) if len(sys.argv) > 1 else 48
RES = [int(v) for v in sys.argv[2:]] or [3]
dev = torch.device("cuda")
L = _lib.lib()
torch.zeros(1, device=dev)

# the HIP runtime this process already uses
hip_path = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip_path = line.split()[-1]
        break
hip = C.CDLL(hip_path)
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
print("HIP runtime:", hip_path, flush=True)

here = os.path.join(ROOT, "tools", "cu_probe")
so = os.path.join(here, "libwhere.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(here, "where.hip")])
W = C.CDLL(so)
W.where_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
W.where_launch.restype = C.c_int


def masked_stream(bits):
    """bits: iterable of mask bit indices (0..255)"""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value), s.value


def where(stream_ptr, n=4096):
    out = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rc = W.where_launch(out.data_ptr(), n, 256, 32768, 200000, stream_ptr)
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype("uint32").reshape(n, 2)
    hw, xcc = o[:, 0], o[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    return sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))


def describe(name, locs):
    per = {}
    for x, se, sh, cu in locs:
        per.setdefault(x, []).append((se, sh, cu))
    print(f"{name}: {len(locs)} distinct CUs; per XCC: " + ", ".join(f"{x}:{len(v)}" for x, v in sorted(per.items())))
    return per


full = where(0)
per = describe("default stream", full)
print("  XCC 0 (se, sh, cu):", per.get(0))

M, d, ld, mlp = UNITS * 8192, 1056, 1088, 2816
torch.manual_seed(0)
a = torch.randn(M, ld, device=dev).bfloat16(); a[:, d:] = 0
w1 = (torch.randn(2 * mlp, ld, device=dev) * 0.03).bfloat16(); w1[:, d:] = 0
w2 = (torch.randn(d, mlp, device=dev) * 0.03).bfloat16()
wq = (torch.randn(3 * d, ld, device=dev) * 0.03).bfloat16(); wq[:, d:] = 0
scale = torch.full((12,), 2.3, device=dev)
hmid = torch.zeros(M, mlp, dtype=torch.bfloat16, device=dev)
yg = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
att = torch.zeros(UNITS, 8192, ld, dtype=torch.bfloat16, device=dev)
y = torch.randn(M, d, device=dev).bfloat16()
hi, lo = ops.split_pair(torch.randn(M, d, device=dev), ld, 8)
gamma, beta = 1 + 0.1 * torch.randn(d, device=dev), 0.1 * torch.randn(d, device=dev)
mod = 0.3 * torch.randn(UNITS, 48 * d, device=dev)[:, 4 * d:6 * d]


def k_w1(): ops.gemm(a[:, :d] if False else a, w1, hmid, epilogue=_lib.EPI_SWIGLU)
def k_w2(): ops.gemm(hmid, w2, yg)
def k_fq(): ops.qkv_attention_fused(a, wq, scale, UNITS, (64, 128), 12, out=att, k=d)
def k_mn(): ops.modnorm_residual_pair(y, hi, lo, gamma, beta, mod, 8192, d)


def timed(stream, fn, reps):
    with torch.cuda.stream(stream):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    return e0, e1


def alone(stream, fn, reps=4, rounds=3):
    ts = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = timed(stream, fn, reps)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


s_def = torch.cuda.current_stream()
base = {}
for nm, fn in (("w1+swiglu", k_w1), ("w2", k_w2), ("fused qkv+attn", k_fq), ("modnorm pair", k_mn)):
    base[nm] = alone(s_def, fn)
    print(f"all CUs, default stream: {nm:16s} {base[nm]*1e3:9.1f} us", flush=True)
mn_bytes = M * d * 8.0

for R in RES:
    # mask bit i: XCC = i % 8, slot j = i // 8 walks (SE fastest, then CU) inside the XCC: reserve slots 0..R-1 of every XCC
    rbits = [i for i in range(256) if i // 8 < R]
    mbits = [i for i in range(256) if i // 8 >= R]
    s_mat, p_mat = masked_stream(mbits)
    s_hbm, p_hbm = masked_stream(rbits)
    pm = describe(f"R={R}: matrix mask", where(p_mat))
    ph = describe(f"R={R}: stream mask", where(p_hbm))
    print("  stream mask, XCC 0 (se, sh, cu):", ph.get(0))
    ncu = 256 - 8 * R
    L.swiftk_set_tuning(2, ncu)
    for nm, fn in (("w1+swiglu", k_w1), ("w2", k_w2), ("fused qkv+attn", k_fq)):
        t = alone(s_mat, fn)
        print(f"R={R}: {ncu} CUs, grid {ncu}: {nm:16s} {t*1e3:9.1f} us  ({t/base[nm]:.3f} x all CUs)", flush=True)
    t_mn = alone(s_hbm, k_mn, reps=2)
    print(f"R={R}: modnorm pair on {8*R} CUs alone: {t_mn*1e3:9.1f} us = {mn_bytes/t_mn/1e9:7.1f} TB/s", flush=True)
    # both at once: the matrix stream runs `reps` launches; the stream stream runs as many norm launches as fit beside them
    for nm, fn in (("w1+swiglu", k_w1), ("w2", k_w2), ("fused qkv+attn", k_fq)):
        reps = 6
        t_alone = alone(s_mat, fn)
        n_mn = max(1, int(reps * t_alone / t_mn + 0.5))
        res = []
        for _ in range(3):
            torch.cuda.synchronize()
            h0, h1 = timed(s_hbm, k_mn, n_mn)
            g0, g1 = timed(s_mat, fn, reps)
            torch.cuda.synchronize()
            res.append((g0.elapsed_time(g1) / reps, h0.elapsed_time(h1) / n_mn))
        res.sort()
        tg, th = res[1]
        print(f"R={R}: together: {nm:16s} {tg*1e3:9.1f} us ({tg/t_alone:.3f} x alone on {ncu} CUs, {tg/base[nm]:.3f} x all CUs)   "
              f"modnorm x{n_mn}: {th*1e3:9.1f} us = {mn_bytes/th/1e9:7.1f} TB/s", flush=True)
    L.swiftk_set_tuning(2, 256)
