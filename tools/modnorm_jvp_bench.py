#!/usr/bin/env python
"""Pair-form ModulatedNorm tangent kernel, a row per wave (tuning key 17 = 0) against blocks walking 32 n rows: python tools/modnorm_jvp_bench.py [units]"""
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
kd = ops.k_pad(BF, d)
torch.manual_seed(1)
y, dy = torch.randn(M, d, device=dev).to(BF), torch.randn(M, d, device=dev).to(BF)
x, dx = torch.randn(M, d, device=dev), torch.randn(M, d, device=dev)
gamma, beta = torch.randn(d, device=dev), torch.randn(d, device=dev)
mod, dmod = torch.randn(B, 2 * d, device=dev) * 0.1, torch.randn(B, 2 * d, device=dev) * 0.1
s = torch.cuda.current_stream().cuda_stream


def state():
    hi = torch.zeros(2 * M, kd, dtype=BF, device=dev)
    lo = torch.empty(2 * M, d, dtype=torch.uint8, device=dev)
    X = torch.cat([x, dx])
    assert L.swiftk_split_pair(X.data_ptr(), d, hi.data_ptr(), kd, lo.data_ptr(), d, 8, 2 * M, d, s) == 0
    return hi, lo


def run(hi, lo):
    assert L.swiftk_modnorm_jvp_pair(y.data_ptr(), dy.data_ptr(), d, hi.data_ptr(), hi.data_ptr() + M * kd * 2, hi.data_ptr(),
                                     hi.data_ptr() + M * kd * 2, kd, lo.data_ptr(), lo.data_ptr() + M * d, gamma.data_ptr(), beta.data_ptr(),
                                     mod.data_ptr(), dmod.data_ptr(), 2 * d, M, d, rps, 1e-6, s) == 0


outs = {}
for key in (0, 1, 2, 4, 0, 1, 2, 4):
    L.swiftk_set_tuning(17, key)
    hi, lo = state()
    run(hi, lo)
    torch.cuda.synchronize()
    if key not in outs:
        outs[key] = (hi.clone(), lo.clone())
    for _ in range(3):
        run(hi, lo)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run(hi, lo)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"key17={key}: {ms * 1e3:.1f} us per call; {M * d * 16 / ms / 1e9:.2f} TB/s over 16 B per element")
for key in (1, 2, 4):
    dh = (outs[key][0].float() - outs[0][0].float()).abs().max().item()
    nl = (outs[key][1] != outs[0][1]).float().mean().item()
    print(f"key17={key} vs 0: max |hi difference| {dh:.3e}, low bytes differing {nl:.2e}")
