#!/usr/bin/env python
"""Calibration only (never on the product path): torch's scaled_dot_product_attention on the window-attention core's shape at the
benchmarked size -- [units x 32 windows, 12 heads, 256 tokens, head_dim 88] bf16, scale 1.0, inputs already window-major (the
product kernels additionally gather / scatter the shifted windows).  python tools/library_attn_ref.py [units]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda")
torch.manual_seed(0)
q, k, v = (torch.randn(B * 32, 12, 256, 88, device=dev).bfloat16() for _ in range(3))
q = F.normalize(q.float(), dim=-1).bfloat16() * 10.0
k = F.normalize(k.float(), dim=-1).bfloat16()
from torch.nn.attention import SDPBackend, sdpa_kernel
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem-efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        with sdpa_kernel(be):
            for _ in range(2):
                o = F.scaled_dot_product_attention(q, k, v, scale=1.0)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    o = F.scaled_dot_product_attention(q, k, v, scale=1.0)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 3)
        t = sorted(ts)[2]
        byts = 4 * q.numel() * 2
        print(f"SDPA {name:14s}: median {t * 1e3:8.1f} us  {byts / t / 1e9:6.2f} TB/s of q+k+v+o  {B * 8.858e9 / t / 1e9:6.1f} TFLOP/s")
    except Exception as ex:  # a backend that does not take head_dim 88
        print(f"SDPA {name:14s}: not available for this shape ({type(ex).__name__}: {str(ex)[:100]})")
