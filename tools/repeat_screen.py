#!/usr/bin/env python
"""Repeatability screen of the round-4 training kernels whose synchronisation is hand-counted: each is launched 40 times on the
same inputs (with a large allocation churned in between, so that timing and addresses of the neighbours change) and every
output is compared bit for bit with the first launch's.  `python tools/repeat_screen.py`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import _lib, ops

L = _lib.lib()
dev = torch.device("cuda", 0)
BF = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
Mh, d, heads, hd, mlp = 8192, 1056, 12, 88, 2816
K = ops.k_pad(BF, d)
a = torch.zeros(2 * Mh, K, dtype=BF, device=dev); a[:, :d] = torch.randn(2 * Mh, d, device=dev).to(BF)
wq = torch.zeros(3 * d, K, dtype=BF, device=dev); wq[:, :d] = (0.03 * torch.randn(3 * d, d, device=dev)).to(BF)
w1 = torch.zeros(2 * mlp, K, dtype=BF, device=dev); w1[:, :d] = (0.05 * torch.randn(2 * mlp, d, device=dev)).to(BF)
scale = torch.log(torch.full((heads,), 10.0, device=dev))
a3, w3 = ops.split3(torch.randn(4096, d, device=dev), 0), ops.split3(0.05 * torch.randn(2 * mlp, d, device=dev), 1)
B, gh, gw = 2, 64, 128
n = gh * gw
pre = torch.nn.functional.normalize(torch.randn(2 * B, n, heads, 3, hd, device=dev), dim=-1).reshape(2 * B, n, -1).to(BF).contiguous()


def qknorm_jvp():
    out = torch.empty(2 * Mh, 3 * d, dtype=BF, device=dev); rn = torch.empty(Mh, 3 * heads, device=dev)
    assert L.swiftk_gemm_jvp(a.data_ptr(), K, wq.data_ptr(), K, out.data_ptr(), 3 * d, Mh, 3 * d, d, _lib.EPI_QKNORM_JVP, scale.data_ptr(),
                             rn.data_ptr(), hd, None, 0, s) == 0
    return out, rn


def swiglu_jvp():
    hm = torch.zeros(2 * Mh, mlp, dtype=BF, device=dev); hp = torch.empty(Mh, 2 * mlp, dtype=BF, device=dev)
    assert L.swiftk_gemm_jvp(a.data_ptr(), K, w1.data_ptr(), K, hp.data_ptr(), 2 * mlp, Mh, 2 * mlp, d, _lib.EPI_SWIGLU_JVP, None, None, 0,
                             hm.data_ptr(), mlp, s) == 0
    return hm, hp


def split3():
    out = torch.zeros(4096, ops.k_pad(BF, 3 * mlp), dtype=BF, device=dev)
    assert L.swiftk_gemm(a3.data_ptr(), a3.stride(0), w3.data_ptr(), w3.stride(0), out.data_ptr(), out.stride(0), 4096, 2 * mlp, a3.shape[1],
                         _lib.BF16, _lib.BF16, _lib.EPI_SWIGLU_SPLIT3, None, None, mlp, s) == 0
    return (out,)


def attn_jvp():
    out = torch.empty(2 * B, n, heads * hd, dtype=BF, device=dev)
    assert L.swiftk_window_attention_jvp(pre.data_ptr(), pre.data_ptr() + B * n * 3 * d * 2, 3 * d, out.data_ptr(),
                                         out.data_ptr() + B * n * d * 2, d, B, gh, gw, heads, hd, 8, 8, _lib.BF16, s) == 0
    return (out,)


bad = 0
for name, fn in (("gemm_jvp qknorm", qknorm_jvp), ("gemm_jvp swiglu", swiglu_jvp), ("swiglu split3", split3), ("attention tangent", attn_jvp)):
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    miss = 0
    for it in range(40):
        junk = torch.empty((it % 7 + 1) * 64 * 1024 * 1024, dtype=torch.uint8, device=dev).fill_(it)  # neighbours move
        got = fn()
        torch.cuda.synchronize()
        miss += sum(0 if torch.equal(g, r) else 1 for g, r in zip(got, ref))
        del junk
    print(f"{name:20s} 40 launches, outputs differing from the first launch: {miss}")
    bad += miss
print("REPEAT SCREEN:", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(0 if bad == 0 else 1)
