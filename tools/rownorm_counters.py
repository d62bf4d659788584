#!/usr/bin/env python
"""One of the wo / w2 GEMM forms at 96 units, launched a few times and nothing else -- the program rocprofv3 --pmc runs for
tools/profile_rownorm_counters.sh.  usage: rownorm_counters.py {rownorm|gemm} {wo|w2} [units] [launches]
  rownorm  swiftk_gemm_modnorm_residual_pair, 64-row workgroups (the complete-row kernel: GEMM + ModulatedNorm in one launch)
  gemm     swiftk_gemm (256 x 352 tiles, ping-pong k-loop) -- the norm kernel is not launched: its share is known (1.28 ms)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
mode, which = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 96
R = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda")
d, ld = 1056, 1088
K, lda = (1056, 1088) if which == "wo" else (2816, 2816)
M = B * 8192
torch.manual_seed(0)
gamma, beta = 1 + 0.1 * torch.randn(d, device=dev), 0.1 * torch.randn(d, device=dev)
mod = 0.3 * torch.randn(B, 48 * d, device=dev)[:, 4 * d:6 * d]
a = torch.randn(M, lda, device=dev).bfloat16(); a[:, K:] = 0
w = (torch.randn(d, lda, device=dev) * 0.03).bfloat16(); w[:, K:] = 0
hi, lo = ops.split_pair(torch.randn(M, d, device=dev), ld, 8)
y = torch.zeros(M, d, dtype=torch.bfloat16, device=dev)
for _ in range(R):
    if mode == "rownorm":
        ops.gemm_modnorm_residual_pair(a, w, hi, lo, gamma, beta, mod, 8192, d, k=K, rows_per_workgroup=64)
    else:
        ops.gemm(a[:, :K] if K % 64 == 0 else a, w[:, :K] if K % 64 == 0 else w, y)
torch.cuda.synchronize()
print(mode, which, "done")
