#!/usr/bin/env python
"""Times swiftk_gemm on the Swift-B shapes (bf16), variants interleaved in one process (A/B rule)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
M = B * 8192
shapes = [("qkv", 3168, 1088, 1056, ops.EPI_NONE), ("wo", 1056, 1088, 1056, ops.EPI_NONE),
          ("w1+swiglu", 5632, 1088, 1056, ops.EPI_SWIGLU), ("w2", 1056, 2816, 2816, ops.EPI_NONE)]
L = _lib.lib()
variants = [(0, 8, 256), (1, 8, 256), (1, 4, 256), (1, 16, 256), (1, 8, 512), (1, 1, 256)]
torch.manual_seed(0)
for name, N, K, Kalg, epi in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    out = torch.empty(M, N // 2 if epi == ops.EPI_SWIGLU else N, dtype=torch.bfloat16, device=dev)
    ref = None
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            L.swiftk_set_tuning(0, v[0]); L.swiftk_set_tuning(1, v[1]); L.swiftk_set_tuning(2, v[2])
            ops.gemm(a, w, out=out, epilogue=epi)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm(a, w, out=out, epilogue=epi)
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 5)
            if rnd == 0:
                if ref is None: ref = out.float().clone()
                else:
                    err = float((out.float() - ref).norm() / ref.norm())
                    assert err < 1e-6, (name, v, err)
    flop = 2.0 * M * N * Kalg
    for v in variants:
        t = sorted(res[v])
        print(f"{name:10s} var={v}  median {t[2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us  {flop/t[2]/1e9:7.1f} TFLOP/s (median)")
