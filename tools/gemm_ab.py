#!/usr/bin/env python
"""A/B of GEMM builds in ONE process (cdna_hip_programming.md rule 24): every lib in swift_amd/csrc/variants/ plus the
product lib runs the Swift-B GEMM shapes in interleaved rounds; outputs are compared bit-for-bit with the first lib's.

  python tools/gemm_ab.py [units] [rounds] [name ...]      (names: product, r01, novm, prio, ...; default: all found)
"""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from swift_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
want = sys.argv[3:]
paths = {"product": os.path.join(ROOT, "swift_amd", "csrc", "libswiftk.so")}
for p in sorted(glob.glob(os.path.join(ROOT, "swift_amd", "csrc", "variants", "libswiftk_*.so"))):
    paths[os.path.basename(p)[len("libswiftk_"):-3]] = p
if want:  # "name", "name@V" (swiftk_set_tuning(0, V): kernel variant), "name+sP" (tuning key 7: start-up stagger, P permille),
    # "name~P" (tuning key 20: ping-pong k-loop on / off)
    paths = {k: paths[k.partition("~")[0].partition("@")[0].partition("+")[0]] for k in want}
libs, variant, stagger, pp = {}, {}, {}, {}
for name, p in paths.items():
    h = C.CDLL(p)
    for fn in ("swiftk_gemm", "swiftk_gemm_qkv_tiled", "swiftk_set_tuning"):
        getattr(h, fn).argtypes, getattr(h, fn).restype = _lib._SIGS[fn]
    libs[name] = h
    pp[name] = name.partition("~")[2]
    name_ = name.partition("~")[0]
    variant[name] = int(name_.partition("@")[2].partition("+")[0] or 1)
    stagger[name] = int(name_.partition("+s")[2] or 0)
names = list(libs)
dev = torch.device("cuda")
M = B * 8192
st = lambda: torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
scale = torch.full((12,), 2.3, device=dev)
shapes = [("qkv_tiled", 3168, 1088, 1056, "tiled"), ("wo", 1056, 1088, 1056, _lib.EPI_NONE),
          ("w1+swiglu", 5632, 1088, 1056, _lib.EPI_SWIGLU), ("w2", 1056, 2816, 2816, _lib.EPI_NONE)]
if os.environ.get("GEMM_AB_TRAIN"):  # the training-only epilogue: d(hidden) = dY W2 through the saved (gate, up) pre-activations
    shapes = [("swiglu_bwd", 2816, 1088, 1056, _lib.EPI_SWIGLU_BWD)]
for sname, N, K, Kalg, epi in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    if K > Kalg:
        a[:, Kalg:] = 0
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    if K > Kalg:
        w[:, Kalg:] = 0
    ncol = N // 2 if epi == _lib.EPI_SWIGLU else (2 * N if epi == _lib.EPI_SWIGLU_BWD else N)
    H = torch.randn(M, 2 * N, device=dev).bfloat16() if epi == _lib.EPI_SWIGLU_BWD else None
    outs = {n: torch.zeros(M, ncol, dtype=torch.bfloat16, device=dev) for n in names}

    def run(n):
        h, o = libs[n], outs[n]
        h.swiftk_set_tuning(0, variant[n])
        h.swiftk_set_tuning(7, stagger[n])
        if pp[n] != "":
            h.swiftk_set_tuning(20, int(pp[n]))
        if epi == "tiled":
            rc = h.swiftk_gemm_qkv_tiled(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), Kalg, scale.data_ptr(), B, 64, 128, 12,
                                         88, 8, 8, st())
        else:
            rc = h.swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), ncol, M, N, Kalg, _lib.BF16, _lib.BF16, epi, None,
                               H.data_ptr() if H is not None else None, 2 * N if H is not None else 0, st())
        assert rc == 0, (n, sname, rc)

    res = {n: [] for n in names}
    for rnd in range(ROUNDS):
        for n in (names if rnd % 2 == 0 else names[::-1]):
            run(n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                run(n)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / 6)
    ref = outs[names[0]]
    flop = 2.0 * M * N * Kalg
    for n in names:
        same = bool(torch.equal(outs[n].view(torch.int16), ref.view(torch.int16)))
        t = sorted(res[n])
        print(f"{sname:10s} {n:10s} median {t[len(t) // 2] * 1e3:8.1f} us  min {t[0] * 1e3:8.1f} us  "
              f"{flop / t[len(t) // 2] / 1e9:7.1f} TFLOP/s  bit-equal-to-{names[0]}: {same}", flush=True)
