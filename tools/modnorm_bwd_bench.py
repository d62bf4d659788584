#!/usr/bin/env python
"""ModulatedNorm backward, one kernel against row pass + column pass (tuning key 16): python tools/modnorm_bwd_bench.py [units]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from swift_amd import _lib, ops

L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rps, d = 8192, 1056
M = B * rps
dev = torch.device("cuda", 0)
BF = torch.bfloat16
y = torch.randn(M, d, device=dev).to(BF)
g = torch.randn(M, d, device=dev)
gamma, beta = torch.randn(d, device=dev), torch.randn(d, device=dev)
mod = torch.randn(B, 2 * d, device=dev) * 0.1
dy = torch.zeros(M, ops.k_pad(BF, d), dtype=BF, device=dev)
dg, db, dm = torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(B, 2 * d, device=dev)
st = torch.empty(2 * M, device=dev)
s = torch.cuda.current_stream().cuda_stream


def run():
    assert L.swiftk_modnorm_bwd(y.data_ptr(), d, g.data_ptr(), dy.data_ptr(), dy.stride(0), gamma.data_ptr(), beta.data_ptr(), mod.data_ptr(),
                                2 * d, dg.data_ptr(), db.data_ptr(), dm.data_ptr(), 2 * d, st.data_ptr(), M, d, rps, 1e-6, _lib.BF16, s) == 0


for key in (0, 1, 2, 4, 8, 0, 1, 2, 4, 8):
    L.swiftk_set_tuning(16, key)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"key16={key}: {ms * 1e3:.1f} us per call; {M * d * 8 / ms / 1e9:.2f} TB/s over 8 B per element (one read of y and g, one write of dy)")
