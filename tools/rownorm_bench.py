#!/usr/bin/env python
"""wo / w2 + ModulatedNorm at small batch: the complete-row kernel (swiftk_gemm_modnorm_residual_pair, 32- and 64-row workgroups)
against what the forward ran before it -- split-K into bf16 slabs + the slab-summing norm (one unit per step) or the tiled GEMM +
the pair norm (beyond) -- interleaved rounds in one process.
usage: rownorm_bench.py [units ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
dev = torch.device("cuda"); L = _lib.lib()
UNITS = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4]
d, ld, mlp = 1056, 1088, 2816
torch.manual_seed(0)
st = lambda: torch.cuda.current_stream().cuda_stream
for B in UNITS:
    M = B * 8192
    gamma, beta = 1 + 0.1 * torch.randn(d, device=dev), 0.1 * torch.randn(d, device=dev)
    mod = 0.3 * torch.randn(B, 48 * d, device=dev)[:, 4 * d:6 * d]
    for nm, K, lda in (("wo", 1056, 1088), ("w2", 2816, 2816)):
        a = torch.randn(M, lda, device=dev).bfloat16(); a[:, K:] = 0
        w = (torch.randn(d, lda, device=dev) * 0.03).bfloat16(); w[:, K:] = 0
        hi, lo = ops.split_pair(torch.randn(M, d, device=dev), ld, 8)
        y = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
        slabs = torch.zeros(2, M, d, dtype=torch.bfloat16, device=dev)
        def rn32(): ops.gemm_modnorm_residual_pair(a, w, hi, lo, gamma, beta, mod, 8192, d, k=K, rows_per_workgroup=32)
        def rn64(): ops.gemm_modnorm_residual_pair(a, w, hi, lo, gamma, beta, mod, 8192, d, k=K, rows_per_workgroup=64)
        def two():
            ops.gemm(a[:, :K] if K % 64 == 0 else a, w[:, :K] if K % 64 == 0 else w, y)
            ops.modnorm_residual_pair(y, hi, lo, gamma, beta, mod, 8192, d)
        def split():
            _lib.check(L.swiftk_gemm_splitk_bf16(a.data_ptr(), lda, w.data_ptr(), lda, slabs.data_ptr(), d, M * d, M, d, K, 2, st()), "splitk")
            ops.modnorm_residual_pair_slabs(slabs, hi, lo, gamma, beta, mod, 8192, d)
        fns = {"complete rows, 32": rn32, "complete rows, 64": rn64, "GEMM + norm": two, "split-K + slab norm": split}
        res = {k: [] for k in fns}
        for f in fns.values(): f()
        torch.cuda.synchronize()
        for rnd in range(7):
            order = list(fns.items())
            for name, fn in (order if rnd % 2 == 0 else order[::-1]):
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): fn()
                e1.record(); torch.cuda.synchronize(); res[name].append(e0.elapsed_time(e1) / 10)
        line = f"units {B} {nm}+norm: "
        for name, t in res.items():
            t = sorted(t); line += f"{name} {t[len(t)//2]*1e3:7.1f} us | "
        print(line, flush=True)
