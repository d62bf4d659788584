#!/usr/bin/env python
"""The packed pair norm in its one-y form against the two-slab form (swiftk_modnorm_residual_pair_halves_bf16) with no, some and all
rows taking the second slab -- same buffers, interleaved rounds.   usage: halves_norm_bench.py [units] [rounds]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import _lib, ops
units = int(sys.argv[1]) if len(sys.argv) > 1 else 12
R = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device("cuda"); L = _lib.lib()
rps, d = 8192, 1056
M = units * rps
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
slabs = torch.randn(2, M, d, generator=g, device=dev).bfloat16()
x = torch.randn(M, d, generator=g, device=dev)
ld = ops.k_pad(torch.bfloat16, d)
hi, lo = ops.split_pair(x, ld, 8)
gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
mod = 0.1 * torch.randn(units, 2 * d, generator=g, device=dev)
tiles = (M // 256) * 3
tail_from = tiles - tiles % 256 if tiles % 256 else tiles
rows_from = (tail_from // 24) * 8 * 256
cases = {"one-y kernel": None,
         "two-slab kernel, no row with a second slab": (ctypes.c_int64 * 3)(M, 1 << 30, 8),
         f"two-slab kernel, slab 1 under tiles >= {tail_from} (rows from {rows_from})": (ctypes.c_int64 * 3)(min(rows_from, M), tail_from, 8),
         "two-slab kernel, every row": "all"}
def run(name):
    t = cases[name]
    if t is None:
        ops.modnorm_residual_pair(slabs[0], hi, lo, gam, bet, mod, rps, d)
    else:
        _lib.check(L.swiftk_modnorm_residual_pair_halves_bf16(slabs.data_ptr(), M * d, None if t == "all" else t, hi.data_ptr(), ld, lo.data_ptr(),
                                                              gam.data_ptr(), bet.data_ptr(), mod.data_ptr(), 2 * d, M, d, rps, 1e-6, st), "halves")
res = {k: [] for k in cases}
for rnd in range(R):
    for k in (list(cases) if rnd % 2 == 0 else list(cases)[::-1]):
        run(k); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(k)
        e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 10)
for k, v in res.items():
    print(f"{k:75s} median {sorted(v)[len(v) // 2] * 1e3:8.1f} us")
