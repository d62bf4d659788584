#!/usr/bin/env python
"""Weight-gradient GEMM (gemm_tn_kernel) with the one-barrier k-loop against the ping-pong k-loop (tuning key 22), the four
shapes of a layer at local batch 8, interleaved rounds in one process; results compared bit for bit.
usage: tn_ab.py [batch] [rounds] [dim]   (dim 1056 = Swift-B: 352-wide tiles; 1280 / 1536 = the larger variants: 320- / 384-wide)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd._lib import lib, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 9
D = int(sys.argv[3]) if len(sys.argv) > 3 else 1056
MLP = (int(8 / 3.0 * D) + 7) // 8 * 8
dev = torch.device("cuda"); L = lib()
Mtok = B * 8192
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
tot = {0: 0.0, 1: 0.0}
for name, rows, cols in (("to_qkv", 3 * D, D), ("wo", D, D), ("w1", 2 * MLP, D), ("w2", D, MLP)):
    ldp, ldq = (rows + 63) // 64 * 64, (cols + 351) // 352 * 352 if cols % 352 == 0 else (cols + 383) // 384 * 384
    dy = torch.zeros(Mtok, ldp, dtype=torch.bfloat16, device=dev); dy[:, :rows] = torch.randn(Mtok, rows, device=dev).bfloat16()
    x = torch.zeros(Mtok, ldq, dtype=torch.bfloat16, device=dev); x[:, :cols] = torch.randn(Mtok, cols, device=dev).bfloat16()
    tiles = ((rows + 255) // 256) * ((cols + 351) // 352)  # (the engine's split rule)
    ks = max(1, min(32, 256 // tiles, Mtok // 64))
    slabs = {p: torch.empty(ks * rows * cols, device=dev) for p in (0, 1)}
    def run(p):
        L.swiftk_set_tuning(22, p)
        check(L.swiftk_gemm_tn_splitk(dy.data_ptr(), ldp, x.data_ptr(), ldq, slabs[p].data_ptr(), cols, rows * cols, rows, cols, Mtok, ks, st), "tn")
    res = {0: [], 1: []}
    for rnd in range(R):
        for p in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            run(p); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(p)
            e1.record(); torch.cuda.synchronize(); res[p].append(e0.elapsed_time(e1) / 10)
    flop = 2.0 * Mtok * rows * cols
    m = {p: sorted(res[p])[R // 2] for p in (0, 1)}
    for p in (0, 1): tot[p] += m[p]
    print(f"{name:7s} ks {ks:2d}: one barrier per k-tile {m[0]*1e3:7.1f} us ({flop/m[0]/1e9:6.0f} TF/s) | ping-pong {m[1]*1e3:7.1f} us ({flop/m[1]/1e9:6.0f} TF/s)"
          f" | bit-equal {torch.equal(slabs[0], slabs[1])}", flush=True)
L.swiftk_set_tuning(22, 1)
print(f"layer: {tot[0]*1e3:.0f} us | {tot[1]*1e3:.0f} us")
