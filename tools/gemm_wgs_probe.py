#!/usr/bin/env python
"""Is the persistent GEMM limited by per-workgroup work or by what the workgroups share (L2, fabric, HBM)?  Same launch with
256 / 128 / 64 / 32 resident workgroups (tuning key 2): intrinsic per-workgroup cost scales time by 2x per halving, contention
on shared resources scales it by less.   usage: gemm_wgs_probe.py [units]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
dev = torch.device("cuda"); L = _lib.lib()
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 96) * 8192
shapes = [("wo", 1056, 1088, 1056, ops.EPI_NONE), ("w1+swiglu", 5632, 1088, 1056, ops.EPI_SWIGLU), ("w2", 1056, 2816, 2816, ops.EPI_NONE)]
for name, N, K, Kalg, epi in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    if K > Kalg:
        a[:, Kalg:] = 0; w[:, Kalg:] = 0
        a, w = a[:, :Kalg], w[:, :Kalg]
    out = torch.empty(M, N // 2 if epi == ops.EPI_SWIGLU else N, dtype=torch.bfloat16, device=dev)
    base = None
    for wgs in ([int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (256, 128, 64, 32, 256)):
        L.swiftk_set_tuning(2, wgs)
        ts = []
        for _ in range(3):
            ops.gemm(a, w, out=out, epilogue=epi); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm(a, w, out=out, epilogue=epi)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
        t = sorted(ts)[1]
        base = base or t
        print(f"{name:10s} wgs={wgs:3d} {t*1e3:9.1f} us  x{t/base:5.2f} of 256-wg time  per-wg rate {2.0*M*N*Kalg/t/1e9/wgs:6.2f} TFLOP/s", flush=True)
L.swiftk_set_tuning(2, 256)
