#!/usr/bin/env python
"""Calibration only (never on the product path): what does the ROCm library GEMM behind torch.matmul (hipBLASLt / rocBLAS) reach
on the Swift-B GEMM shapes at the benchmarked size?  bf16 operands, fp32 accumulate, plain bf16 output, no fused epilogue.
python tools/library_gemm_ref.py [units]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda")
M = B * 8192
torch.manual_seed(0)
for name, N, K in (("to_qkv", 3168, 1056), ("wo", 1056, 1056), ("w1 (plain, 2 x the SwiGLU output bytes)", 5632, 1056), ("w2", 1056, 2816)):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            torch.matmul(a, w.t(), out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    t = sorted(ts)[2]
    print(f"{name:42s} M {M} N {N} K {K}: median {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s")
    del a, w, out
