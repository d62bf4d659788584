#!/usr/bin/env python
"""Where the complete-row kernel's time goes at one unit per step: timing ablations (tuning key 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
dev = torch.device("cuda"); L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 32
d, ld = 1056, 1088
M = B * 8192
torch.manual_seed(0)
gamma, beta = 1 + 0.1 * torch.randn(d, device=dev), 0.1 * torch.randn(d, device=dev)
mod = 0.3 * torch.randn(B, 48 * d, device=dev)[:, 4 * d:6 * d]
for nm, K, lda in (("wo", 1056, 1088), ("w2", 2816, 2816)):
    a = torch.randn(M, lda, device=dev).bfloat16(); a[:, K:] = 0
    w = (torch.randn(d, lda, device=dev) * 0.03).bfloat16(); w[:, K:] = 0
    hi, lo = ops.split_pair(torch.randn(M, d, device=dev), ld, 8)
    def run(): ops.gemm_modnorm_residual_pair(a, w, hi, lo, gamma, beta, mod, 8192, d, k=K, rows_per_workgroup=rows)
    out = []
    for dbg in (0, 1, 2, 3):
        L.swiftk_set_tuning(24, dbg)
        ts = []
        for _ in range(5):
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
        out.append(f"dbg {dbg}: {sorted(ts)[2]*1e3:6.1f} us")
    L.swiftk_set_tuning(24, 0)
    print(f"{nm} ({rows} rows, {B} units): " + " | ".join(out) + "   (1 = no row phase, 2 = one k-tile, 3 = both)", flush=True)
