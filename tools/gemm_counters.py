#!/usr/bin/env python
"""One GEMM flavour, launched a few times and nothing else -- the program rocprofv3 --pmc runs for tools/profile_gemm_counters.sh.
usage: gemm_counters.py {w1swiglu|w2|wo} {0|1: ping-pong k-loop, tuning key 20} [units] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
shape, pp = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 96
R = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda"); L = _lib.lib()
N, K, Kalg, epi = {"w1swiglu": (5632, 1088, 1056, ops.EPI_SWIGLU), "w2": (1056, 2816, 2816, ops.EPI_NONE), "wo": (1056, 1088, 1056, ops.EPI_NONE)}[shape]
M = B * 8192
torch.manual_seed(0)
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
if K > Kalg:
    a[:, Kalg:] = 0; w[:, Kalg:] = 0
    a, w = a[:, :Kalg], w[:, :Kalg]
out = torch.empty(M, N // 2 if epi == ops.EPI_SWIGLU else N, dtype=torch.bfloat16, device=dev)
L.swiftk_set_tuning(20, pp)
for _ in range(R):
    ops.gemm(a, w, out=out, epilogue=epi)
torch.cuda.synchronize()
print(shape, "pp", pp, "done")
