#!/usr/bin/env python
"""Race screen of the persistent attention backward (QK-norm backward inside): 60 launches on the same inputs, shifted and
unshifted windows alternating, outputs compared bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd import _lib
dev = torch.device("cuda"); L = _lib.lib()
B, gh, gw, heads, hd = 8, 64, 128, 12, 88
n = gh * gw
torch.manual_seed(0)
pre = torch.nn.functional.normalize(torch.randn(B, n, heads, 3, hd, device=dev), dim=-1)
pre[..., 0, :] *= 10.0
pre = pre.reshape(B, n, -1).bfloat16()
o = torch.randn(B, n, 1088, device=dev).bfloat16() * 0.1
do = torch.randn(B, n, 1088, device=dev).bfloat16()
scale = torch.log(torch.full((heads,), 10.0, device=dev))
rn = torch.rand(B * n, 3 * heads, device=dev) + 0.5
st = torch.cuda.current_stream().cuda_stream
ref = None
bad = 0
for it in range(60):
    out = torch.zeros(B, n, 3200, dtype=torch.bfloat16, device=dev)
    ds = torch.zeros(heads, device=dev)
    assert L.swiftk_window_attention_bwd_qknorm(pre.data_ptr(), 3168, o.data_ptr(), do.data_ptr(), 1088, out.data_ptr(), 3200, scale.data_ptr(),
                                                rn.data_ptr(), ds.data_ptr(), B, gh, gw, heads, hd, 8 * (it & 1), 8 * (it & 1), _lib.BF16, st) == 0
    torch.cuda.synchronize()
    key = it & 1
    if ref is None: ref = {}
    if key not in ref: ref[key] = out.clone()
    elif not torch.equal(ref[key], out): bad += 1
print("runs 60, mismatching outputs:", bad)
