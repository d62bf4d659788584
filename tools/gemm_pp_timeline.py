#!/usr/bin/env python
"""Ping-pong GEMM diagnostics (lib built with -DSWIFTK_X_PP=.. -DSWIFTK_PP_STAMP=1, selected with SWIFTK_LIB):
 (1) ablation timings, tuning key 3: 0 = as shipped, 8 = every stage re-reads k-tile 0, 1 = no DMA, 4 = no epilogue, 5 = neither;
 (2) s_memtime stamps around every barrier of one tile (bit 64): mean cycles a wave of group 0 / group 1 spends in each MEM
     phase, at the barrier behind it, in each COMPUTE phase, at the barrier behind that.
usage: SWIFTK_LIB=.../libswiftk_ppstamp.so python tools/gemm_pp_timeline.py [units]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from swift_amd import _lib, ops

L = _lib.lib()
dev = torch.device("cuda")
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 96) * 8192
st = lambda: torch.cuda.current_stream().cuda_stream
shapes = [("w1 plain", 5632, 1088, 1056, ops.EPI_NONE), ("w1+swiglu", 5632, 1088, 1056, ops.EPI_SWIGLU), ("w2", 1056, 2816, 2816, ops.EPI_NONE)]
for name, N, K, Kalg, epi in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    if K > Kalg:
        a[:, Kalg:] = 0
        w[:, Kalg:] = 0
        a, w = a[:, :Kalg], w[:, :Kalg]
    out = torch.empty(M, N // 2 if epi == ops.EPI_SWIGLU else N, dtype=torch.bfloat16, device=dev)
    for dbg in (0, 0, 8, 1, 4, 5):
        L.swiftk_set_tuning(3, dbg)
        ts = []
        for _ in range(4):
            ops.gemm(a, w, out=out, epilogue=epi)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm(a, w, out=out, epilogue=epi)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        t = sorted(ts)[1]
        print(f"{name:10s} dbg={dbg} {t*1e3:8.1f} us  {2.0*M*N*Kalg/t/1e9:7.1f} TFLOP/s", flush=True)
    L.swiftk_set_tuning(3, 0)
    if epi != ops.EPI_NONE:
        continue
    for dbg in (64, 64 + 8, 64 + 1):
        log = torch.zeros(8 * 2 * 20 * 16, dtype=torch.int64, device=dev)
        L.swiftk_set_tuning(3, dbg)
        _lib.check(L.swiftk_gemm(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), N, M, N, Kalg, _lib.BF16, _lib.BF16,
                                 _lib.EPI_NONE, None, log.data_ptr(), 0, st()), "gemm")
        torch.cuda.synchronize()
        L.swiftk_set_tuning(3, 0)
        t = log.cpu().numpy().reshape(8, 2, 20, 16).astype(np.float64)
        nkt = int((t[0, 0, :, 0] > 0).sum())
        for grp in range(2):
            x = t[:, grp, 1:nkt - 1, :]  # full k-tiles in the middle of the tile
            prev_end = t[:, grp, 0:nkt - 2, 15]
            mem = [x[:, :, 0] - prev_end] + [x[:, :, 4 * p] - x[:, :, 4 * p - 1] for p in range(1, 4)]
            b1 = [x[:, :, 4 * p + 1] - x[:, :, 4 * p] for p in range(4)]
            cmp_ = [x[:, :, 4 * p + 2] - x[:, :, 4 * p + 1] for p in range(4)]
            b2 = [x[:, :, 4 * p + 3] - x[:, :, 4 * p + 2] for p in range(4)]
            tot = (x[:, :, 15] - prev_end).mean()
            f = lambda v: " ".join(f"{q.mean():6.0f}" for q in v)
            print(f"{name} dbg={dbg} group {grp}: k-tile {tot:7.0f} cycles ({nkt} k-tiles);  MEM {f(mem)} | barrier {f(b1)} | COMPUTE {f(cmp_)} | barrier {f(b2)}", flush=True)
