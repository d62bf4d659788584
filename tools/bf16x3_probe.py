#!/usr/bin/env python
"""Probe: fp32-grade GEMM out of three bf16 products through the ordinary bf16 kernel (K-concatenation):
  A ~ A_hi + A_lo, W ~ W_hi + W_lo (bf16 each);  A W^T ~ [A_hi | A_lo | A_hi] [W_hi | W_hi | W_lo]^T   (lo x lo dropped)
against the exact-fp32 MFMA GEMM of the parity engine: speed and error vs an fp64 product.  python tools/bf16x3_probe.py [units]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
M = B * 8192
torch.manual_seed(0)
def split(x):
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return hi, lo
def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
    return sorted(ts)[2]
for name, N, K in (("to_qkv", 3168, 1056), ("wo", 1056, 1056), ("w1", 5632, 1056), ("w2", 1056, 2816)):
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.03
    ah, al = split(a); wh, wl = split(w)
    kp = ops.k_pad(torch.bfloat16, 3 * K)
    a3 = torch.zeros(M, kp, dtype=torch.bfloat16, device=dev); w3 = torch.zeros(N, kp, dtype=torch.bfloat16, device=dev)
    a3[:, :K], a3[:, K:2 * K], a3[:, 2 * K:3 * K] = ah, al, ah
    w3[:, :K], w3[:, K:2 * K], w3[:, 2 * K:3 * K] = wh, wh, wl
    kf = ops.k_pad(torch.float32, K)
    af = torch.zeros(M, kf, device=dev); wf = torch.zeros(N, kf, device=dev); af[:, :K] = a; wf[:, :K] = w
    o3 = torch.empty(M, N, device=dev); of = torch.empty(M, N, device=dev)
    t3 = timeit(lambda: ops.gemm(a3, w3, out=o3, out_dtype=torch.float32))
    tf = timeit(lambda: ops.gemm(af, wf, out=of, out_dtype=torch.float32))
    ob = ops.gemm(a3[:, :ops.k_pad(torch.bfloat16, K)].contiguous() if False else torch.nn.functional.pad(ah, (0, ops.k_pad(torch.bfloat16, K) - K)),
                  torch.nn.functional.pad(wh, (0, ops.k_pad(torch.bfloat16, K) - K)), out_dtype=torch.float32)
    rows = slice(0, 2048)
    ref = a[rows].double() @ w.double().t()
    rel = lambda x: float((x[rows].double() - ref).norm() / ref.norm())
    print(f"{name:7s} N {N} K {K}: bf16x3 {t3 * 1e3:8.1f} us (rel err {rel(o3):.2e})   fp32 MFMA {tf * 1e3:8.1f} us (rel err {rel(of):.2e})   "
          f"plain bf16 rel err {rel(ob):.2e}   speed-up {tf / t3:.2f}x")
