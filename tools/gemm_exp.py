#!/usr/bin/env python
"""GEMM structure experiments (tuning key 3, lib built with -DSWIFTK_GEMM_INSTR=1): which of DMA / barrier / epilogue bounds the
persistent loop.  dbg bits: 1 = no DMA in the loop, 2 = no barrier, 4 = no epilogue, 8 = every stage re-reads k-tile 0 (the DMA is
issued as usual but hits L2) -- all give wrong results; timing only.   usage: gemm_exp.py [units]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
dev = torch.device("cuda"); L = _lib.lib()
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 8) * 8192
shapes = [("qkv+qknorm", 3168, 1088, 1056, ops.EPI_QKNORM), ("w1 plain", 5632, 1088, 1056, ops.EPI_NONE), ("wo", 1056, 1088, 1056, ops.EPI_NONE),
          ("w1+swiglu", 5632, 1088, 1056, ops.EPI_SWIGLU), ("w2", 1056, 2816, 2816, ops.EPI_NONE)]
scale = torch.full((12,), 2.3, device=dev)
for name, N, K, Kalg, epi in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    if K > Kalg:  # K = 16.5 k-tiles: zero pad columns, views of the valid width (the GEMM then skips the padded half k-tile)
        a[:, Kalg:] = 0; w[:, Kalg:] = 0
        a, w = a[:, :Kalg], w[:, :Kalg]
    out = torch.empty(M, N // 2 if epi == ops.EPI_SWIGLU else N, dtype=torch.bfloat16, device=dev)
    bias = scale if epi == ops.EPI_QKNORM else None
    for dbg in (0, 0, 8, 12, 4, 1):
        L.swiftk_set_tuning(3, dbg)
        ts = []
        for _ in range(4):
            ops.gemm(a, w, out=out, epilogue=epi, bias=bias); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm(a, w, out=out, epilogue=epi, bias=bias)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
        t = sorted(ts)[1]
        print(f"{name:10s} dbg={dbg} {t*1e3:8.1f} us  {2.0*M*N*Kalg/t/1e9:7.1f} TFLOP/s")
L.swiftk_set_tuning(3, 0)
