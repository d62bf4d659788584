#!/usr/bin/env python
"""Headline benchmark: 6 h forecast sample-steps / second of the Swift-B sCM 1-step sampler
(BASELINE.json configs[1]: 128x256x69 synthetic ERA5-shaped fields, bf16 GEMM operands).

  python bench.py --gpus N --steps K --warmup W [--batch B] [--dtype bf16|f32]

A "step" is one pass of the hot path over one batch of B (member, IC) units on every rank: draw
the latent noise, one fused Swift-B network evaluation (patch gather .. un-patchify + sCM update),
the residual state update in physical units and re-standardisation -- i.e. one iteration of the
reference's rollout loop (generate.py:97-131) with state, forcings and outputs resident in HBM.
value = N * B * K / t (whole job), t = max over ranks of the barrier-bracketed wall time.

Multi-GPU: units are independent, so ranks shard them with no data-path collective ("weak"
scaling: B per rank fixed); RCCL carries the one-time weight broadcast from rank 0 and the
barriers only.

Extra legs printed in the same JSON line:
  roofline      the dominant kernel (w1 GEMM + fused SwiGLU, 42.7 % of all FLOPs): algorithmic FLOPs per
                launch / mean launch time from HIP events recorded on the launch stream during the timed region
  cpu_baseline  (rank 0, N == 1) the CPU oracle -- a plain-PyTorch fp32 restatement of the reference, pinned
                to it by tests/golden -- timed on the host cores for a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SWIFT_B = dict(window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=12, dim=1056, heads=12)
IMG, NV, NF = (128, 256), 69, 3
FLOP_PER_EVAL = 2.7535e12  # 2*MACs of one Swift-B network evaluation
PEAK_BF16, PEAK_F32 = 2.5e15, 157.3e12  # MI355X_MICROARCH.md: dense MFMA peaks


def build_net(dev, rank, world):
    from swift_amd.models.precond import PassPrecond
    from swift_amd.utils.detinit import swinv2_state

    mcfg = dict(_target_="swift.models.swinv2.SwinV2", **SWIFT_B)
    net = PassPrecond(mcfg, img_resolution=list(IMG), img_channels=NV, condition_channels=NV + NF, auxiliary_dim=1)
    state = None
    if rank == 0:
        state = swinv2_state(grid=(64, 128), in_channels=2 * NV + NF, out_channels=NV, patch_size=(2, 2), depth=12, dim=1056,
                             heads=12, seed=1234)
        net.load_state_dict(state)
    net = net.to(dev).eval()
    if world > 1:  # weights travel once over RCCL/xGMI (north_star: broadcast for weights)
        for p in net.parameters():
            dist.broadcast(p.data, src=0)
    return net, state


def cpu_baseline(state, sample_steps: int):
    """The oracle on the host cores: `sample_steps` 1-member x 1-IC x 1-step forecasts (BASELINE config 1)."""
    from oracle import sampler as osamp
    from oracle.swinv2 import OracleNet, SwinCfg
    from swift_amd.utils.detinit import det_normal

    cfg = SwinCfg(img_resolution=IMG, in_channels=2 * NV + NF, out_channels=NV, window_size=(16, 16), shift_size=(8, 8),
                  patch_size=(2, 2), depth=12, dim=1056, heads=12, auxiliary_dim=1)
    onet = OracleNet(cfg, state, NV, NV + NF)
    cond, lat = det_normal((1, NV + NF, *IMG), 1, "cond"), det_normal((1, NV, *IMG), 1, "lat")
    run = lambda: osamp.scm_solver(onet, lat, cond, 0.6, num_steps=1, sigma_min=0.02, sigma_max=200.0)
    run()  # warm-up (first call pays allocator / oneDNN primitive creation)
    t0 = time.perf_counter()
    for _ in range(sample_steps):
        run()
    dt = time.perf_counter() - t0
    return dict(value=sample_steps / dt, unit="sample-steps/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{sample_steps} x (1 member x 1 IC x 1 step), Swift-B scm 1-step, fp32, after 1 warm-up; "
                       f"{dt / sample_steps:.2f} s per sample-step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="(member, IC) units per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--cpu-steps", type=int, default=2, help="sample-steps of the CPU baseline (0 = skip)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one HIP graph (pays off below ~8 units per step)")
    ap.add_argument("--solver", default="scm", choices=["scm", "2s", "dpm"],
                    help="scm = BASELINE configs[1] (default, the metric's workload); 2s / dpm = configs[2], multi-step ODE sampler")
    ap.add_argument("--num-steps", type=int, default=None, help="solver steps (default: 1 for scm, 20 for 2s, 8 for dpm)")
    a = ap.parse_args()

    from swift_amd import _lib, dist as sdist, ops
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import RolloutEngine, unit_seed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = sdist.setup_torch()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = sdist.get_torch_device()
    lib = _lib.lib()
    for kv in filter(None, os.environ.get("SWIFTK_TUNE", "").split(",")):  # kernel A/B knobs, e.g. "3:8" (swiftk_set_tuning)
        lib.swiftk_set_tuning(*(int(x) for x in kv.split(":")))
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    B, K, W = a.batch, a.steps, a.warmup

    net, state = build_net(dev, rank, world)
    ds = SyntheticERA5Dataset([f"v{i}" for i in range(NV)], [f"f{i}" for i in range(NF)], img_resolution=IMG, length=64,
                              seed=1234)
    nsteps = a.num_steps or {"scm": 1, "2s": 20, "dpm": 8}[a.solver]
    evals = {"scm": nsteps, "2s": 2 * nsteps - 1, "dpm": nsteps}[a.solver]  # network evaluations per sample-step
    eng = RolloutEngine(net, ds, interval=6, solver=a.solver, denoise_dtype=dtype, num_steps=nsteps)
    # this rank's units: contiguous block of the flattened (member, IC) space
    units = [(u // 64, u % 64) for u in range(rank * B, rank * B + B)]
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    X = torch.randn(B, NV, *IMG, generator=g, device=dev)
    forc = torch.randn(1, B, NF, *IMG, generator=g, device=dev)  # one staged forcing slab, reused every step
    mx, sx, st = eng.stats(dev)
    gens = [torch.Generator(device=dev).manual_seed(unit_seed(m, ic)) for m, ic in units]
    phys = torch.empty_like(X)
    z = torch.empty_like(X)

    def step():
        for b, gg in enumerate(gens):
            z[b].normal_(generator=gg)
        Y = eng.sampler((X, forc[0]), latents=z)
        ops.rollout_update(X, Y, mx, sx, st, phys=phys)

    if a.graph:  # noise stays outside the graph (per-unit generators); everything else of the step is one replay
        graph = eng.capture_step(X, forc[0], z, phys)

        def step():  # noqa: F811
            for b, gg in enumerate(gens):
                z[b].normal_(generator=gg)
            graph.replay()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(W):
        step()
    mlp2 = 2 * int(8 / 3.0 * 1056)
    if not a.graph:  # the per-launch event pairs of the roofline leg cannot be recorded inside a replayed graph
        lib.swiftk_profile_gemm(_lib.EPI_SWIGLU, mlp2)
    sync()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    sync()
    dt = time.perf_counter() - t0
    tot_ms, n_launch = ctypes.c_double(0), ctypes.c_int64(0)
    lib.swiftk_profile_collect(ctypes.byref(tot_ms), ctypes.byref(n_launch))
    lib.swiftk_profile_gemm(-1, 0)
    if not torch.isfinite(phys).all():
        raise SystemExit("non-finite forecast state")
    # second roofline leg, outside the timed region: the window-attention kernel (north_star's named kernel) over two
    # more steps, HIP events on its launch stream
    att_ms, att_n = ctypes.c_double(0), ctypes.c_int64(0)
    if a.dtype == "bf16" and not a.graph:
        lib.swiftk_profile_gemm(_lib.PROF_ATTENTION, 0)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        lib.swiftk_profile_collect(ctypes.byref(att_ms), ctypes.byref(att_n))
        lib.swiftk_profile_gemm(-1, 0)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        # per-launch fabric traffic of the two roofline kernels from the committed PMC passes (valid for the default workload)
        traffic = {}
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as f:
                pt = json.load(f)
            if pt.get("units_per_step") == B and a.dtype == "bf16":
                traffic = pt
        except OSError:
            pass
        value = world * B * K / dt
        M = B * 64 * 128
        flop_launch = 2.0 * M * mlp2 * 1056  # algorithmic: K = 1056, not the padded 1088
        avg_s = (tot_ms.value / max(n_launch.value, 1)) * 1e-3
        peak = PEAK_BF16 if a.dtype == "bf16" else PEAK_F32
        ach = flop_launch / avg_s if avg_s > 0 else 0.0
        line = {
            "metric": "6h forecast steps/sec (members x ICs) on 128x256x69 ERA5",
            "value": value,
            "unit": "sample-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * dt / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": ("Swift-B sCM 1-step sampler, 128x256x69 (BASELINE configs[1]): noise + fused network "
                                    "eval + residual state update per step") if a.solver == "scm" and nsteps == 1 else
                       f"Swift-B {a.solver} sampler, num_steps {nsteps} = {evals} network evaluations per sample-step, 128x256x69 "
                       "(BASELINE configs[2])", "units_per_gpu_per_step": B, "hip_graph": bool(a.graph),
                       "params": 225980976, "parallelism": f"units sharded over {world} GPU(s), no data-path collective"},
            "e2e": {"tflops": FLOP_PER_EVAL * evals * value / 1e12,
                    "frac_of_dense_mfma_peak": FLOP_PER_EVAL * evals * value / (peak * world)},
            "roofline": {"kernel": "gemm_kernel<bf16,bf16,SWIGLU> (w1 + SwiGLU)" if a.dtype == "bf16" else
                         "gemm_kernel<f32,f32,SWIGLU> (w1 + SwiGLU)", "bound": "mfma", "achieved": ach / 1e12,
                         "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic.get("gemm_swiglu"),
                         "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json)",
                         "launches": int(n_launch.value), "avg_launch_ms": avg_s * 1e3,
                         "flop_per_launch": flop_launch},
        }
        if att_n.value > 0:
            # per sample-layer the kernel reads qkv (8192 x 3168 bf16) and writes 8192 x 1056 bf16
            # = 69.2 MB, for 8.858 GFLOP of QK^T + PV: HBM-bound in bf16 (ridge 312 flop/B > 128 flop/B)
            att_s = att_ms.value / att_n.value * 1e-3
            att_bytes, att_flop = B * 8192 * 4 * 1056 * 2.0, B * 8.858e9
            line["attention_roofline"] = {
                "kernel": "attn_pipe_kernel (shifted-window attention, bf16, window-tiled q/k/v)", "bound": "hbm",
                "achieved": att_bytes / att_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": att_bytes / att_s / 8e12,
                "traffic": traffic.get("attention"), "launches": int(att_n.value), "avg_launch_ms": att_s * 1e3, "bytes_per_launch": att_bytes,
                "mfma_tflops": att_flop / att_s / 1e12, "mfma_frac": att_flop / att_s / PEAK_BF16}
        if world == 1 and a.cpu_steps > 0:
            line["cpu_baseline"] = cpu_baseline(state, a.cpu_steps)
            line["vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
