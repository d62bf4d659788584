#!/usr/bin/env python
"""to_qkv + window attention of one Swift-B layer: the two-kernel path (swiftk_gemm_qkv_tiled + swiftk_window_attention, q/k/v
window-tiled through HBM) against swiftk_qkv_attention_fused, interleaved rounds in one process, random operands.
usage: qkv_attn_ab.py [units] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
dev = torch.device("cuda"); L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
grid, heads, hd, d, K = (64, 128), 12, 88, 1056, 1088
M = B * grid[0] * grid[1]
torch.manual_seed(0)
a = torch.randn(M, K, device=dev).bfloat16(); a[:, d:] = 0
w = (torch.randn(3 * d, K, device=dev) * 0.03).bfloat16(); w[:, d:] = 0
scale = torch.log(torch.tensor([10.0] * 11 + [100.0], device=dev))
qkv = torch.empty(B, 32, heads, 3, 256, hd, dtype=torch.bfloat16, device=dev)
o2 = torch.zeros(B, 8192, K, dtype=torch.bfloat16, device=dev)
of = torch.zeros(B, 8192, K, dtype=torch.bfloat16, device=dev)
def two():
    ops.gemm_qkv_tiled(a, w, scale, B, grid, heads, (8, 8), out=qkv, k=d)
    ops.window_attention_tiled(qkv, scale, grid, heads, (8, 8), out=o2)
def gemm_only():
    ops.gemm_qkv_tiled(a, w, scale, B, grid, heads, (8, 8), out=qkv, k=d)
def fused():
    ops.qkv_attention_fused(a, w, scale, B, grid, heads, (8, 8), out=of, k=d)
def fused_noattn():
    L.swiftk_set_tuning(4, 1 << 8); fused(); L.swiftk_set_tuning(4, 0)
def fused_w0():
    L.swiftk_set_tuning(4, 2 << 8); fused(); L.swiftk_set_tuning(4, 0)
def fused_w0_noattn():
    L.swiftk_set_tuning(4, 3 << 8); fused(); L.swiftk_set_tuning(4, 0)
def fused_pp():
    L.swiftk_set_tuning(21, 1); fused(); L.swiftk_set_tuning(21, 0)
def stag(n):
    def f():
        L.swiftk_set_tuning(4, (n << 2) << 8); fused(); L.swiftk_set_tuning(4, 0)
    return f
fns = {"two kernels": two, "fused, waves 4-7 +192 cyc": stag(2), "  (to_qkv GEMM alone)": gemm_only, "fused": fused, "fused, ping-pong k-loop (tuning key 21)": fused_pp, "  (fused, attention core skipped)": fused_noattn,
       "  (fused, every item reads head 0's weights: WRONG results)": fused_w0, "  (the same, attention core skipped)": fused_w0_noattn}
res = {k: [] for k in fns}
for rnd in range(R):
    for k in (list(fns) if rnd % 2 == 0 else list(fns)[::-1]):
        fns[k](); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): fns[k]()
        e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 4)
ofp = torch.zeros_like(of)
L.swiftk_set_tuning(21, 1); ops.qkv_attention_fused(a, w, scale, B, grid, heads, (8, 8), out=ofp, k=d); L.swiftk_set_tuning(21, 0)
two(); fused(); torch.cuda.synchronize()
print("ping-pong k-loop output bit-equal to the one-barrier loop's:", bool(torch.equal(ofp, of)))
rel = float((of[..., :d].float() - o2[..., :d].float()).norm() / o2[..., :d].float().norm())
flop = 2.0 * M * 3 * d * d + B * 8.858e9
for k in fns:
    t = sorted(res[k]); med = t[len(t) // 2]
    print(f"{k:60s} median {med*1e3:8.1f} us  min {t[0]*1e3:8.1f} us   {flop/med/1e9:7.1f} TFLOP/s (qkv + attention FLOPs)", flush=True)
print(f"fused vs two-kernel output: rel-L2 {rel:.2e}")
