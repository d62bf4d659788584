#!/usr/bin/env python
"""Where does the exact-fp32 engine lose accuracy against the CPU's fp32?  
The oracle (plain PyTorch-CPU fp32) runs a depth-D Swift-B-width network with ONE op family at a time replaced by the HIP
kernel (fp32 operands), and every variant is measured against the oracle in fp64:
  cpu        every op on the CPU (the reference's arithmetic)
  +gemm      the large Linears on swiftk_gemm (fp32 MFMA 16x16x4 chain)
  +attn      cosine window attention on swiftk_window_attention (fp32)
  +norm      ModulatedNorm on swiftk_modnorm_residual
  engine     the whole fp32 engine
usage: python tests/fp32_bisect.py [depth]   (lives under tests/: it uses the oracle as its yardstick)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import oracle.swinv2 as osw
from oracle.swinv2 import OracleNet, SwinCfg
from swift_amd import ops
from swift_amd.models.precond import PassPrecond
from swift_amd.utils.detinit import det_normal, swinv2_state

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
img, nv, nf, dim, heads = (64, 64), 69, 3, 1056, 12
state = swinv2_state(grid=(32, 32), in_channels=2 * nv + nf, out_channels=nv, patch_size=(2, 2), depth=depth, dim=dim, heads=heads, seed=5)
cfg = SwinCfg(img_resolution=img, in_channels=2 * nv + nf, out_channels=nv, window_size=(16, 16), shift_size=(8, 8), patch_size=(2, 2),
              depth=depth, dim=dim, heads=heads, auxiliary_dim=1)
x, cond, t = det_normal((2, nv, *img), 5, "x"), det_normal((2, nv + nf, *img), 5, "cond"), torch.tensor([0.4, 1.3])
onet = OracleNet(cfg, state, nv, nv + nf)
o64 = OracleNet(cfg, {k: v.double() for k, v in state.items()}, nv, nv + nf)
with torch.no_grad():
    truth = o64(x.double(), t.double(), cond.double(), torch.tensor(0.6, dtype=torch.float64))
rel = lambda a: float((a.double().cpu() - truth).norm() / truth.norm())

lin0, att0, norm0 = F.linear, osw.cosine_window_attention, osw.modulated_norm
wcache = {}
def hip_linear(a, w, b=None):
    if w.shape[0] < 64 or a.numel() // a.shape[-1] < 256:
        return lin0(a, w, b)
    K = w.shape[1]
    kp = ops.k_pad(torch.float32, K)
    if id(w) not in wcache:
        wcache[id(w)] = ops.pad_cols(w.to(dev), kp, torch.float32)
    a2 = ops.pad_cols(a.reshape(-1, K).to(dev), kp, torch.float32)
    y = (ops.gemm_chunked(a2, wcache[id(w)], chunk_k=CHUNK) if CHUNK else ops.gemm(a2, wcache[id(w)], out_dtype=torch.float32))
    y = y.cpu().reshape(*a.shape[:-1], w.shape[0])
    return y if b is None else y + b
def hip_attn(qkv, scale, heads_, naive=False, emulate_bf16=False):
    Bw = qkv.shape[0]
    out = ops.window_attention(qkv.to(dev).contiguous(), scale.reshape(-1).to(dev), (16, 16), heads_, (0, 0))
    return out.cpu().float().reshape(Bw, 256, -1)
def hip_norm(xx, t_lat, p, prefix, eps=1e-6):
    B, n, d = xx.shape
    mod = lin0(t_lat, p[prefix + "modulation.weight"], p[prefix + "modulation.bias"]).to(dev)
    acc = torch.zeros(B * n, d, device=dev)
    ops.modnorm_residual(xx.reshape(B * n, d).to(dev).contiguous(), acc, p[prefix + "norm.weight"].to(dev), p[prefix + "norm.bias"].to(dev), mod, n)
    return acc.cpu().reshape(B, n, d)
def run(lin=lin0, att=att0, norm=norm0):
    F.linear, osw.cosine_window_attention, osw.modulated_norm = lin, att, norm
    try:
        with torch.no_grad():
            return onet(x, t, cond, 0.6)
    finally:
        F.linear, osw.cosine_window_attention, osw.modulated_norm = lin0, att0, norm0
CHUNK = int(os.environ.get("SWIFTK_CHUNK", "256"))
CK = CHUNK
res = {"cpu": rel(run()), "+gemm (chains of 256 k)": rel(run(lin=hip_linear))}
CHUNK = 0
res.update({"+gemm (one chain)": rel(run(lin=hip_linear)), "+attn": rel(run(att=hip_attn)), "+norm": rel(run(norm=hip_norm)),
       "+gemm+attn+norm": rel(run(hip_linear, hip_attn, hip_norm))})
mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=depth, dim=dim, heads=heads)
net = PassPrecond(mcfg, img_resolution=list(img), img_channels=nv, condition_channels=nv + nf, auxiliary_dim=1)
net.load_state_dict(state)
net = net.to(dev).eval()
from swift_amd import _lib
with torch.no_grad():
    res["engine (chains of 256 k)"] = rel(net(x.to(dev), t.to(dev), cond.to(dev), 0.6))
    _lib.lib().swiftk_set_tuning(13, 0)
    res["engine (one chain)"] = rel(net(x.to(dev), t.to(dev), cond.to(dev), 0.6))
    _lib.lib().swiftk_set_tuning(13, 256)
    if CK != 256:
        _lib.lib().swiftk_set_tuning(13, CK)
        res[f"engine (chains of {CK} k)"] = rel(net(x.to(dev), t.to(dev), cond.to(dev), 0.6))
        _lib.lib().swiftk_set_tuning(13, 256)
print(f"depth {depth}, rel-L2 against the fp64 oracle:")
for k, v in res.items():
    print(f"  {k:26s} {v:.3e}  ({v / res['cpu']:.2f} x cpu)")
