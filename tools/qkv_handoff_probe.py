#!/usr/bin/env python
"""Fused to_qkv + window attention (swiftk_qkv_attention_fused, ping-pong k-loop): what the k-loop -> attention-core hand-off
costs.  Timing probes through tuning key 4 (bits 8..; results are WRONG while set): 1 = attention core skipped, 4 = the
hand-off without its 135 KB of ds_write_b64 per item, 8 = without the norm arithmetic as well.
usage: qkv_handoff_probe.py [units] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
dev = torch.device("cuda"); L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
R = int(sys.argv[2]) if len(sys.argv) > 2 else 7
grid, heads, hd, d, K = (64, 128), 12, 88, 1056, 1088
M = B * grid[0] * grid[1]
torch.manual_seed(0)
a = torch.randn(M, K, device=dev).bfloat16(); a[:, d:] = 0
w = (torch.randn(3 * d, K, device=dev) * 0.03).bfloat16(); w[:, d:] = 0
scale = torch.log(torch.tensor([10.0] * 11 + [100.0], device=dev))
of = torch.zeros(B, 8192, K, dtype=torch.bfloat16, device=dev)
def probe(bits):
    def f():
        L.swiftk_set_tuning(4, bits << 8); ops.qkv_attention_fused(a, w, scale, B, grid, heads, (8, 8), out=of, k=d); L.swiftk_set_tuning(4, 0)
    return f
fns = {"shipped kernel": probe(0), "no parking writes (4)": probe(4), "no norm, no parking writes (8)": probe(8),
       "attention core skipped (1)": probe(1), "core skipped, no parking writes (5)": probe(5), "core skipped, no norm / writes (9)": probe(9)}
res = {k: [] for k in fns}
for rnd in range(R):
    for k in (list(fns) if rnd % 2 == 0 else list(fns)[::-1]):
        fns[k](); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): fns[k]()
        e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 4)
flop = 2.0 * M * 3 * d * d + B * 8.858e9
base = None
for k in fns:
    t = sorted(res[k]); med = t[len(t) // 2]
    base = base or med
    print(f"{k:42s} median {med*1e3:8.1f} us  min {t[0]*1e3:8.1f} us  {100*(med/base-1):+6.2f} %   ({B} units)", flush=True)
