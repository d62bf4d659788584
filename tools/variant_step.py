"""One of the reference's larger Swift variants (era5-swinv2-1.4-scm.yaml:29-36) stepping through the bf16 engine, for
`rocprofv3 --kernel-trace --stats -- python3 tools/variant_step.py --dim 1280 --heads 16` (per-kernel shares of a variant)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=1280)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--units", type=int, default=32)
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    import torch

    import bench
    from swift_amd.generating.factory import sampler_factory
    from swift_amd.models.precond import PassPrecond
    from swift_amd.utils.detinit import swinv2_state

    dev = torch.device("cuda:0")
    NV, NF, IMG = bench.NV, bench.NF, bench.IMG
    mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=a.depth,
                dim=a.dim, heads=a.heads)
    net = PassPrecond(mcfg, img_resolution=list(IMG), img_channels=NV, condition_channels=NV + NF, auxiliary_dim=1)
    net.load_state_dict(swinv2_state(grid=(64, 128), in_channels=2 * NV + NF, out_channels=NV, patch_size=(2, 2), depth=a.depth,
                                     dim=a.dim, heads=a.heads, seed=7))
    net = net.to(dev).eval()
    sampler = sampler_factory("scm", net, denoise_dtype=torch.bfloat16, num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    g = torch.Generator(device=dev).manual_seed(0)
    cond = torch.randn(a.units, NV + NF, *IMG, generator=g, device=dev)
    with torch.no_grad():
        for _ in range(2):
            sampler(cond)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            sampler(cond)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"dim {a.dim} heads {a.heads} depth {a.depth} units {a.units}: {1e3 * dt:.2f} ms per step, {a.units / dt:.1f} sample-steps/s")


if __name__ == "__main__":
    main()
