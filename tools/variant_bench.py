#!/usr/bin/env python
"""Larger Swift variants of configs/experiment/era5-swinv2-1.4-scm.yaml:21-36 on the same kernels:
1-step sCM forecast throughput of the bf16 inference engine, per variant.   python tools/variant_bench.py [units] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from swift_amd.generating.factory import sampler_factory
from swift_amd.models.precond import PassPrecond
from swift_amd.utils.detinit import swinv2_state

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
NV, NF, IMG = 69, 3, (128, 256)
VARIANTS = [("Swift-B 226M", 1056, 12, 12), ("468M", 1280, 16, 16), ("664M", 1536, 16, 16)]  # (name, dim, heads, depth)


def flops_per_eval(dim, heads, depth):
    ntok, mlp = 64 * 128, int(8 / 3 * dim)
    layer = 2 * ntok * (dim * 3 * dim + dim * dim + dim * 2 * mlp + mlp * dim) + 4 * ntok * 256 * dim + 2 * 2 * dim * 2 * dim
    return 2 * ntok * 564 * dim + depth * layer + 2 * ntok * dim * 276


for name, dim, heads, depth in VARIANTS:
    mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=depth, dim=dim,
                heads=heads)
    net = PassPrecond(mcfg, img_resolution=list(IMG), img_channels=NV, condition_channels=NV + NF, auxiliary_dim=1)
    net.load_state_dict(swinv2_state(grid=(64, 128), in_channels=2 * NV + NF, out_channels=NV, patch_size=(2, 2), depth=depth, dim=dim,
                                     heads=heads, seed=7))
    net = net.to(dev).eval()
    sampler = sampler_factory("scm", net, denoise_dtype=torch.bfloat16, num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    g = torch.Generator(device=dev).manual_seed(0)
    cond = torch.randn(B, NV + NF, *IMG, generator=g, device=dev)
    with torch.no_grad():
        for _ in range(2):
            sampler(cond)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            out = sampler(cond)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    fl = flops_per_eval(dim, heads, depth) * B
    print(json.dumps({"variant": name, "dim": dim, "heads": heads, "head_dim": dim // heads, "depth": depth, "units": B,
                      "sample_steps_per_s": B / dt, "ms_per_step": 1e3 * dt, "tflops": fl / dt / 1e12, "frac_of_bf16_peak": fl / dt / 2.5e15,
                      "finite": bool(torch.isfinite(out).all())}), flush=True)
    del net, sampler
    torch.cuda.empty_cache()
