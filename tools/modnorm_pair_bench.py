#!/usr/bin/env python
"""ModulatedNorm + residual at the bench shape (units x 8192 rows x 1056): the fp32-stream kernel (12 B per element: y 2, x 4 in;
x 4, bf16 copy 2 out) against the pair kernel (10 B: y, hi, lo in; hi, lo out), interleaved rounds in one process.
usage: modnorm_pair_bench.py [units] [rounds]      (SWIFTK_LIB selects an A/B build of the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import ops, _lib
dev = torch.device("cuda"); L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
R = int(sys.argv[2]) if len(sys.argv) > 2 else 7
M, d, ld = B * 8192, 1056, 1088
torch.manual_seed(0)
y = torch.randn(M, d, device=dev).bfloat16()
x0 = torch.randn(M, d, device=dev)
gamma, beta = 1 + 0.1 * torch.randn(d, device=dev), 0.1 * torch.randn(d, device=dev)
mod = 0.3 * torch.randn(B, 48 * d, device=dev)[:, 4 * d:6 * d]
xc = torch.zeros(M, ld, dtype=torch.bfloat16, device=dev)
hi, lo = ops.split_pair(x0, ld, 16)
hi8, lo8 = ops.split_pair(x0, ld, 8)
x = x0.clone()
def f32(): ops.modnorm_residual(y, x, gamma, beta, mod, 8192, xcopy=xc)
def pair(): ops.modnorm_residual_pair(y, hi, lo, gamma, beta, mod, 8192, d)
def pair8(): ops.modnorm_residual_pair(y, hi8, lo8, gamma, beta, mod, 8192, d)
hi8r, lo8r = ops.split_pair(x0, ld, 8)
def pair8_rows():  # the row-per-wave form of the 8-bit kernel (tuning key 6 bit 2)
    L.swiftk_set_tuning(6, 7); ops.modnorm_residual_pair(y, hi8r, lo8r, gamma, beta, mod, 8192, d); L.swiftk_set_tuning(6, 3)
def variant(bits):
    h_, l_ = ops.split_pair(x0, ld, 8)
    def f():
        L.swiftk_set_tuning(6, 3 | bits); ops.modnorm_residual_pair(y, h_, l_, gamma, beta, mod, 8192, d); L.swiftk_set_tuning(6, 3)
    return f
extra = {}
f32(); pair(); pair8(); pair8_rows(); torch.cuda.synchronize()
for nm, (h_, l_) in (("bf16 low part", (hi, lo)), ("8-bit low part", (hi8, lo8))):
    print(f"after one call, {nm}: hi == bf16 copy of the fp32 stream on", float((h_[:, :d] == xc[:, :d]).float().mean()),
          "of the elements; rel-L2 of the pair's value vs fp32 stream", float((ops.pair_value(h_, l_, d) - x).norm() / x.norm()))
res = {"fp32 stream (12 B/elt)": [], "pair, bf16 lo (10 B/elt)": [], "pair, 8-bit lo, packed (8 B/elt)": [], "pair, 8-bit lo, row per wave (8 B/elt)": []}
res.update({k: [] for k in extra})
for rnd in range(R):
    order = list(zip(res, (f32, pair, pair8, pair8_rows))) + list(extra.items())
    for name, fn in (order if rnd % 2 == 0 else order[::-1]):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize(); res[name].append(e0.elapsed_time(e1) / 5)
for (name, t), bpe in zip(res.items(), (12.0, 10.0, 8.0, 8.0)):
    t = sorted(t); med = t[len(t) // 2]
    print(f"{name}: median {med*1e3:8.1f} us  min {t[0]*1e3:8.1f} us  {M*d*bpe/med/1e6:7.1f} GB/s of algorithmic bytes", flush=True)
