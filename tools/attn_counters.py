#!/usr/bin/env python
"""One variant of the to_qkv + window attention kernel, launched a few times and nothing else -- the program rocprofv3 --pmc
runs for tools/profile_attn_counters.sh.  usage: attn_counters.py {fused|fused_nocore|gemm} [units] [launches]
  fused         swiftk_qkv_attention_fused as shipped
  fused_nocore  the same kernel with its attention core skipped (tuning key 4 bit 8: the to_qkv k-loop, norm and stores remain)
  gemm          swiftk_gemm_qkv_tiled alone (the plain persistent GEMM with the QK-norm epilogue)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
mode = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
R = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda"); L = _lib.lib()
grid, heads, hd, d, K = (64, 128), 12, 88, 1056, 1088
M = B * grid[0] * grid[1]
torch.manual_seed(0)
a = torch.randn(M, K, device=dev).bfloat16(); a[:, d:] = 0
w = (torch.randn(3 * d, K, device=dev) * 0.03).bfloat16(); w[:, d:] = 0
scale = torch.log(torch.tensor([10.0] * 11 + [100.0], device=dev))
out = torch.zeros(B, 8192, K, dtype=torch.bfloat16, device=dev)
qkv = torch.empty(B, 32, heads, 3, 256, hd, dtype=torch.bfloat16, device=dev) if mode == "gemm" else None
if mode == "fused_nocore":
    L.swiftk_set_tuning(4, 1 << 8)
for _ in range(R):
    if mode == "gemm":
        ops.gemm_qkv_tiled(a, w, scale, B, grid, heads, (8, 8), out=qkv, k=d)
    else:
        ops.qkv_attention_fused(a, w, scale, B, grid, heads, (8, 8), out=out, k=d)
torch.cuda.synchronize()
print(mode, "done")
