#!/usr/bin/env python
"""Weight-gradient GEMM shapes of one Swift-B layer (local batch 8): split-K GEMM + slab reduce + the two operand
transposes, each timed alone.   python tools/wgrad_bench.py [units] [ks overrides ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from swift_amd import _lib
from swift_amd._lib import lib, check, BF16

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
Mtok = B * 8192
st = lambda: torch.cuda.current_stream().cuda_stream

def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [("to_qkv", 3168, 1056), ("wo", 1056, 1056), ("w1", 5632, 1056), ("w2", 1056, 2816)]
tot = 0.0
for name, rows, cols in shapes:
    dy = torch.randn(Mtok, rows, device=dev).bfloat16()
    x = torch.randn(Mtok, cols, device=dev).bfloat16()
    dy_t = torch.empty(rows, Mtok, dtype=torch.bfloat16, device=dev)
    x_t = torch.empty(cols, Mtok, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(rows, cols, device=dev)
    tiles = ((rows + 255) // 256) * ((cols + 351) // 352)
    ks0 = max(1, min(32, 512 // tiles, Mtok // 64))
    t_tr = timeit(lambda: (check(lib().swiftk_transpose(dy.data_ptr(), rows, dy_t.data_ptr(), Mtok, Mtok, rows, BF16, st()), "t"),
                           check(lib().swiftk_transpose(x.data_ptr(), cols, x_t.data_ptr(), Mtok, Mtok, cols, BF16, st()), "t")))
    flop = 2.0 * Mtok * rows * cols
    line = f"{name:7s} {rows}x{cols} tiles {tiles:3d}: transposes {t_tr:7.1f} us;"
    best = None
    for ks in sorted({ks0, 256 // tiles, max(1, 256 // tiles) * 2, 768 // tiles, 1024 // tiles} | {int(a) for a in sys.argv[2:]}):
        if ks < 1 or ks > 64:
            continue
        slabs = torch.empty(ks * rows * cols, device=dev)
        t_g = timeit(lambda: check(lib().swiftk_gemm_splitk(dy_t.data_ptr(), Mtok, x_t.data_ptr(), Mtok, slabs.data_ptr(), cols, rows * cols,
                                                            rows, cols, Mtok, BF16, ks, st()), "g"))
        t_r = timeit(lambda: check(lib().swiftk_reduce_slabs(slabs.data_ptr(), cols, rows * cols, ks, out.data_ptr(), cols, rows, cols, 1,
                                                             st()), "r"))
        line += f"  ks={ks}{'*' if ks == ks0 else ''}: {t_g:6.1f}+{t_r:5.1f} us ({flop / t_g / 1e6:5.0f} TF/s)"
        if ks == ks0:
            best = t_g + t_r
    tot += best + t_tr
    print(line, flush=True)
    # TN form: no transposes
    line = f"{name:7s} TN form:"
    ref = torch.empty(rows, cols, device=dev)
    slabs = torch.empty(ks0 * rows * cols, device=dev)
    check(lib().swiftk_gemm_splitk(dy_t.data_ptr(), Mtok, x_t.data_ptr(), Mtok, slabs.data_ptr(), cols, rows * cols, rows, cols, Mtok, BF16,
                                   ks0, st()), "g")
    check(lib().swiftk_reduce_slabs(slabs.data_ptr(), cols, rows * cols, ks0, ref.data_ptr(), cols, rows, cols, 0, st()), "r")
    ldp, ldq = (rows + 63) // 64 * 64, (cols + 351) // 352 * 352
    dyp = torch.zeros(Mtok, ldp, dtype=torch.bfloat16, device=dev); dyp[:, :rows] = dy
    xp = torch.zeros(Mtok, ldq, dtype=torch.bfloat16, device=dev); xp[:, :cols] = x
    for ks in sorted({ks0, max(1, 256 // tiles)} | {int(a) for a in sys.argv[2:]}):
        slabs = torch.empty(ks * rows * cols, device=dev)
        got = torch.empty(rows, cols, device=dev)
        f = lambda: check(lib().swiftk_gemm_tn_splitk(dyp.data_ptr(), ldp, xp.data_ptr(), ldq, slabs.data_ptr(), cols, rows * cols, rows, cols,
                                                      Mtok, ks, st()), "tn")
        t_g = timeit(f)
        check(lib().swiftk_reduce_slabs(slabs.data_ptr(), cols, rows * cols, ks, got.data_ptr(), cols, rows, cols, 0, st()), "r")
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        line += f"  ks={ks}: {t_g:6.1f} us ({flop / t_g / 1e6:5.0f} TF/s) max err vs NT {err:.1e}{' BIT-EQUAL' if ks == ks0 and torch.equal(got, ref) else ''}"
    print(line, flush=True)
print(f"layer total (shipped ks): {tot:.0f} us = {2.0 * Mtok * 1056 * 12672 / tot / 1e6:.0f} TFLOP/s incl. transposes")
