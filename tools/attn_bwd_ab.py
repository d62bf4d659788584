#!/usr/bin/env python
"""Attention backward at the training shape (local batch x 32 windows x 12 heads): persistent LDS-DMA kernel (tuning key 9 = 1)
against the one-workgroup-per-item kernel (0), interleaved rounds.   usage: attn_bwd_ab.py [batch] [rounds] [head_dim] [heads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import _lib
dev = torch.device("cuda"); L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
hd = int(sys.argv[3]) if len(sys.argv) > 3 else 88
heads = int(sys.argv[4]) if len(sys.argv) > 4 else 12
gh, gw = 64, 128
W3, D = 3 * heads * hd, (heads * hd + 63) // 64 * 64
n = gh * gw
torch.manual_seed(0)
pre = torch.nn.functional.normalize(torch.randn(B, n, heads, 3, hd, device=dev), dim=-1)
pre[..., 0, :] *= 10.0
pre = pre.reshape(B, n, -1).bfloat16()
o = torch.randn(B, n, D, device=dev).bfloat16() * 0.1
do = torch.randn(B, n, D, device=dev).bfloat16()
outs = {k: torch.zeros_like(pre) for k in (0, 1)}
st = lambda: torch.cuda.current_stream().cuda_stream
scale = torch.log(torch.full((heads,), 10.0, device=dev))
def run(k, scaled=False):
    L.swiftk_set_tuning(9, k)
    rc = L.swiftk_window_attention_bwd_scaled(pre.data_ptr(), W3, o.data_ptr(), do.data_ptr(), D, outs[k].data_ptr(), W3,
                                              scale.data_ptr() if scaled else None, B, gh, gw, heads, hd, 8, 8, _lib.BF16, st())
    assert rc == 0
res = {0: [], 1: []}
for rnd in range(R):
    for k in ((0, 1) if rnd % 2 == 0 else (1, 0)):
        run(k); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(k)
        e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 5)
L.swiftk_set_tuning(9, 1)
for dbg, name in ((8, "no max sweep"), (1, "no pass-A sweep 2"), (9, "no pass A at all"), (2, "no pass-B loop"), (4, "no output stores"), (15, "images + fragment rows only")):
    L.swiftk_set_tuning(4, dbg << 16); run(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run(1)
    e1.record(); torch.cuda.synchronize()
    print(f"  persistent, {name:28s} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us")
L.swiftk_set_tuning(4, 0)
run(1, True); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run(1, True)
e1.record(); torch.cuda.synchronize()
print(f"persistent, logit scales given (max-free)      {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us")
run(1)
flop = B * 32 * heads * 5 * 2.0 * 256 * 256 * hd
for k, name in ((0, "one workgroup per item"), (1, "persistent, LDS-DMA images")):
    t = sorted(res[k]); med = t[len(t) // 2]
    print(f"{name:30s} median {med*1e3:8.1f} us  min {t[0]*1e3:8.1f} us  {flop/med/1e9:6.1f} TFLOP/s (5 products)")
print("rel-L2 between the two:", float((outs[1].float() - outs[0].float()).norm() / outs[0].float().norm()))
# A/B of single scheduling choices of the persistent kernel (tuning key 4, bits 16..), logit scales given, interleaved rounds
ab = {"shipped": 0, "Q image requested at the item's top (round 3)": 16}
rab = {k: [] for k in ab}
for rnd in range(R + 2):
    for k in (list(ab) if rnd % 2 == 0 else list(ab)[::-1]):
        L.swiftk_set_tuning(4, ab[k] << 16); run(1, True); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(1, True)
        e1.record(); torch.cuda.synchronize(); rab[k].append(e0.elapsed_time(e1) / 10)
L.swiftk_set_tuning(4, 0)
for k, t in rab.items():
    t = sorted(t); print(f"A/B {k:48s} median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us")
