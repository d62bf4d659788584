#!/usr/bin/env python
"""ModulatedNorm + residual at the bench shape (units x 8192 rows x 1056): variants by tuning key 6 (bit 0: non-temporal residual
stream, bit 1: chunked kernel), interleaved rounds in one process, results compared with variant 1.
usage: modnorm_bench.py [units] [rounds] [key-6 values ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
dev = torch.device("cuda"); L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variants = [int(v) for v in sys.argv[3:]] or [1, 3, 2, 0]
M, d, ld = B * 8192, 1056, 1088
torch.manual_seed(0)
y = torch.randn(M, d, device=dev).bfloat16()
x0 = torch.randn(M, d, device=dev)
gamma, beta = 1 + 0.1 * torch.randn(d, device=dev), 0.1 * torch.randn(d, device=dev)
mod = 0.3 * torch.randn(B, 48 * d, device=dev)[:, 4 * d:6 * d]
xc = torch.zeros(M, ld, dtype=torch.bfloat16, device=dev)
bytes_ = M * d * 12.0
res, outs = {v: [] for v in variants}, {}
for v in variants:
    L.swiftk_set_tuning(6, v)
    x = x0.clone(); ops.modnorm_residual(y, x, gamma, beta, mod, 8192, xcopy=xc); outs[v] = (x, xc.clone())
x = x0.clone()
for rnd in range(R):
    for v in (variants if rnd % 2 == 0 else variants[::-1]):
        L.swiftk_set_tuning(6, v)
        ops.modnorm_residual(y, x, gamma, beta, mod, 8192, xcopy=xc); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.modnorm_residual(y, x, gamma, beta, mod, 8192, xcopy=xc)
        e1.record(); torch.cuda.synchronize(); res[v].append(e0.elapsed_time(e1) / 5)
ref = outs[variants[0]]
for v in variants:
    t = sorted(res[v]); med = t[len(t) // 2]
    dx = float((outs[v][0] - ref[0]).abs().max()); same = bool(torch.equal(outs[v][1], ref[1]))
    print(f"key6={v}: median {med*1e3:8.1f} us  min {t[0]*1e3:8.1f} us  {bytes_/med/1e9:7.1f} GB/s  max|dx| vs key6={variants[0]}: {dx:.2e}  bf16 copy equal: {same}", flush=True)
L.swiftk_set_tuning(6, 3)
