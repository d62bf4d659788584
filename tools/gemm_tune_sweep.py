import os, sys
sys.path.insert(0, os.getcwd())
import torch
from swift_amd import ops, _lib
L = _lib.lib(); dev = torch.device("cuda")
M = 96 * 8192
shapes = [("w1+swiglu", 5632, 1088, 1056, ops.EPI_SWIGLU), ("w2", 1056, 2816, 2816, ops.EPI_NONE), ("wo", 1056, 1088, 1056, ops.EPI_NONE)]
for name, N, K, Kalg, epi in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    if K > Kalg:
        a[:, Kalg:] = 0; w[:, Kalg:] = 0; a, w = a[:, :Kalg], w[:, :Kalg]
    out = torch.empty(M, N // 2 if epi == ops.EPI_SWIGLU else N, dtype=torch.bfloat16, device=dev)
    res = {}
    cfgs = [(1, 8), (1, 4), (1, 16), (1, 6), (1, 12), (2, 248), (2, 240)]
    for rnd in range(5):
        for key, val in cfgs:
            L.swiftk_set_tuning(1, 8); L.swiftk_set_tuning(2, 256)
            L.swiftk_set_tuning(key, val)
            ops.gemm(a, w, out=out, epilogue=epi); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm(a, w, out=out, epilogue=epi)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((key, val), []).append(e0.elapsed_time(e1) / 5)
    L.swiftk_set_tuning(1, 8); L.swiftk_set_tuning(2, 256)
    print(name, {f"key{k}={v}": round(sorted(t)[len(t)//2] * 1e3, 1) for (k, v), t in res.items()})
