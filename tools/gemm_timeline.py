"""Per-tile timeline of the persistent GEMM (tuning key 3 bit 32): s_memtime stamps of wave 0 in every 32nd workgroup at
[first k-step top | last MFMA issued | epilogue barrier passed | stores issued | next first k-step top]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from swift_amd import _lib, ops


def main():
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    M = 65536
    for name, N, K in (("wo", 1056, 1088), ("qkv", 3168, 1088), ("w1-plain", 5632, 1088)):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        log = torch.zeros(8 * 64 * 8, dtype=torch.int64, device=dev)
        for _ in range(2):
            ops.gemm(a, w, out=c)
        lib.swiftk_set_tuning(3, 32)
        _lib.check(lib.swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, c.data_ptr(), N, M, N, K, _lib.BF16, _lib.BF16, _lib.EPI_NONE, None,
                                   log.data_ptr(), 0, torch.cuda.current_stream().cuda_stream), "gemm")
        torch.cuda.synchronize()
        lib.swiftk_set_tuning(3, 0)
        t = log.cpu().numpy().reshape(8, 64, 8).astype(np.float64)
        ntile = int((t[0, :, 0] > 0).sum())
        clk = 1.0  # ticks (the counter's rate is not assumed: the shares below are what matter)
        rows = []
        for wg in range(8):
            for i in range(ntile - 1):
                x = t[wg, i]
                rows.append([(x[1] - x[0]) / clk, (x[2] - x[1]) / clk, (x[3] - x[2]) / clk, (x[4] - x[3]) / clk, (x[4] - x[0]) / clk, x[5] / clk, x[6] / clk])
        r = np.array(rows)
        tot = r[:, 4].mean()
        print(f"{name:9s} N={N}: tiles/WG {ntile}; share of a tile's time (mean over {len(r)} tiles): k-loop {100 * r[:,0].mean() / tot:5.1f} %  "
              f"waiting at the epilogue barrier {100 * r[:,1].mean() / tot:4.1f} %  epilogue (convert, LDS transpose, store issue) "
              f"{100 * r[:,2].mean() / tot:4.1f} %  store drain + barrier to the next k-step {100 * r[:,3].mean() / tot:4.1f} %;  inside the "
              f"k-loop: waiting for own DMA {100 * r[:,5].mean() / tot:4.1f} %, at the k-step barrier {100 * r[:,6].mean() / tot:4.1f} %")


main()
