#!/usr/bin/env python
"""One weight-gradient GEMM (gemm_tn_kernel), launched a few times and nothing else -- the program rocprofv3 --pmc runs for
tools/profile_tn_counters.sh.   usage: tn_counters.py {qkv|wo|w1|w2} {0|1: ping-pong k-loop, tuning key 22} [batch] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import _lib
from swift_amd._lib import lib, check
shape, pp = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
R = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda"); L = lib()
rows, cols = {"qkv": (3168, 1056), "wo": (1056, 1056), "w1": (5632, 1056), "w2": (1056, 2816)}[shape]
Mtok = B * 8192
torch.manual_seed(0)
ldp, ldq = (rows + 63) // 64 * 64, (cols + 351) // 352 * 352
dy = torch.zeros(Mtok, ldp, dtype=torch.bfloat16, device=dev); dy[:, :rows] = torch.randn(Mtok, rows, device=dev).bfloat16()
x = torch.zeros(Mtok, ldq, dtype=torch.bfloat16, device=dev); x[:, :cols] = torch.randn(Mtok, cols, device=dev).bfloat16()
tiles = ((rows + 255) // 256) * ((cols + 351) // 352)
ks = max(1, min(32, 256 // tiles, Mtok // 64))
slabs = torch.empty(ks * rows * cols, device=dev)
L.swiftk_set_tuning(22, pp)
st = torch.cuda.current_stream().cuda_stream
for _ in range(R):
    check(L.swiftk_gemm_tn_splitk(dy.data_ptr(), ldp, x.data_ptr(), ldq, slabs.data_ptr(), cols, rows * cols, rows, cols, Mtok, ks, st), "tn")
torch.cuda.synchronize()
print(shape, "pp", pp, "ks", ks, "done")
