"""to_qkv GEMM timing at the Swift-B layer shape: row-major QK-norm epilogue vs the window-tiled store."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from swift_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    heads, gh, gw = 12, 64, 128
    M, K = B * gh * gw, 1088
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    a[:, 1056:] = 0
    w = (torch.randn(3 * heads * 88, K, device=dev) * 0.03).to(torch.bfloat16)
    scale = torch.full((heads,), 2.3, device=dev)
    c = torch.empty(M, 3 * heads * 88, device=dev, dtype=torch.bfloat16)
    ct = torch.empty(B, 32, heads, 3, 256, 88, device=dev, dtype=torch.bfloat16)
    flop = 2.0 * M * 3168 * 1056
    for name, fn in (("plain  ", lambda: ops.gemm(a, w, out=c)),
                     ("qknorm ", lambda: ops.gemm(a, w, out=c, epilogue=ops.EPI_QKNORM, bias=scale)),
                     ("tiled00", lambda: ops.gemm_qkv_tiled(a, w, scale, B, (gh, gw), heads, (0, 0), out=ct, k=1056)),
                     ("tiled88", lambda: ops.gemm_qkv_tiled(a, w, scale, B, (gh, gw), heads, (8, 8), out=ct, k=1056))):
        us = timeit(fn)
        print(f"{name}: {us:7.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)


main()
