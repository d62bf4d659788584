#!/usr/bin/env python
"""Does an HBM-bound kernel co-run with the persistent GEMM?  Stream A: w1+SwiGLU GEMM x R; stream B: an elementwise pass over
`mb` MB x R2.  Prints each alone and both together (wall time between events on a third, joining stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from swift_amd import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
M, N, K = B * 8192, 5632, 1088
a = torch.randn(M, K, device=dev).bfloat16(); a[:, 1056:] = 0
w = (torch.randn(N, K, device=dev) * 0.03).bfloat16(); w[:, 1056:] = 0
o = torch.zeros(M, N // 2, dtype=torch.bfloat16, device=dev)
x = torch.randn(M, 1056, device=dev)
y = torch.randn(M, 1056, device=dev).bfloat16()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
R = 10

def gemm():
    ops.gemm(a, w, o, epilogue=_lib.EPI_SWIGLU)

def ew():
    x.add_(y)       # 2 + 4 + 4 B per element

def timed(fa, fb, ra, rb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_event(e0); sb.wait_event(e0)
    with torch.cuda.stream(sa):
        for _ in range(ra):
            fa()
        ea.record()
    with torch.cuda.stream(sb):
        for _ in range(rb):
            fb()
        eb.record()
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(ea), e0.elapsed_time(eb), e0.elapsed_time(e1)

for _ in range(2):
    timed(gemm, ew, R, R)
ta = timed(gemm, ew, R, 0)
tb = timed(gemm, ew, 0, R)
print(f"GEMM alone   : {ta[0] / R * 1e3:8.1f} us/launch")
print(f"ew alone     : {tb[1] / R * 1e3:8.1f} us/launch  ({x.numel() * 10 / (tb[1] / R * 1e-3) / 1e12:.2f} TB/s)")
for rb in (R // 2, R, 2 * R, 4 * R):
    t = timed(gemm, ew, R, rb)
    print(f"both R={R} rb={rb:3d}: gemm stream {t[0]:7.2f} ms, ew stream {t[1]:7.2f} ms, total {t[2]:7.2f} ms; serial would be "
          f"{ta[0] + tb[1] / R * rb:7.2f} ms")
