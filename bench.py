#!/usr/bin/env python
"""Headline benchmark: 6 h forecast sample-steps / second of the Swift-B sCM 1-step sampler
(BASELINE.json configs[1]: 128x256x69 synthetic ERA5-shaped fields, bf16 GEMM operands).

  python bench.py --gpus N --steps K --warmup W [--batch B] [--dtype bf16|f32]
  python bench.py --gpus N --rollout 12x64x60           (BASELINE configs[3]: the north-star rollout, strong scaling)

A "step" is one pass of the hot path over one batch of B (member, IC) units on every rank: draw the latent noise, one
fused Swift-B network evaluation (patch gather .. un-patchify + sCM update), the residual state update in physical
units and re-standardisation -- one iteration of the reference's rollout loop (generate.py:97-131) with state,
forcings and outputs resident in HBM -- followed by the step's output collection: a fixed-order fp64 checksum per
unit, all-gathered over RCCL (8 bytes per unit instead of the 9 MB state).
value = N * B * K / t (whole job), t = max over ranks of the barrier-bracketed wall time.  The default B = 96 is the
per-GPU share of configs[3] (12 members x 64 ICs on 8 GPUs = 12 members x 8 ICs each).

Multi-GPU: units are independent, so ranks shard the flattened IC-major (IC, member) space in contiguous blocks with no
data-path collective on the state ("weak" scaling: B per rank fixed); RCCL carries the one-time weight broadcast, the
per-step checksum all-gather and the barriers.  Launched bare (`python bench.py --gpus 8`, no WORLD_SIZE) the script
starts the N ranks itself, before anything touches the GPU; under torch.distributed.run it is one of the ranks.  It
exits non-zero when the process group's world size is not --gpus.

Extra objects in the same JSON line:
  roofline        dominant kernel (w1 GEMM + fused SwiGLU, 42.7 % of all FLOPs): algorithmic FLOPs per launch / mean launch
                  time from HIP events recorded on the launch stream during the timed region
  attention_roofline  the fused to_qkv + window-attention kernel against the MFMA roofline over its fused FLOPs (two more
                  steps, outside the timed region)
  rccl, checksum  the process group that ran and what it collected
  parity_engine   (N == 1) the exact-fp32 engine -- the configuration that meets the 1e-4 tolerance -- on the same
                  workload: sample-steps/s, fraction of the fp32 matrix peak, its attention kernel's MFMA fraction
  bf16_vs_fp32    (N == 1) relative L2 distance of the bf16 engine's state from the fp32 engine's after 1, 10 and 60
                  autoregressive steps on the same units and noise
  cpu_baseline    (N == 1) the CPU oracle -- a plain-PyTorch fp32 restatement of the reference, pinned to it by
                  tests/golden -- timed on the host cores for a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SWIFT_B = dict(window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=12, dim=1056, heads=12)
IMG, NV, NF = (128, 256), 69, 3
MEMBERS = 12  # ensemble members per initial condition (BASELINE configs[3]); units are IC-major: u -> (u // 12, u % 12)
FLOP_PER_EVAL = 2.7535e12  # 2*MACs of one Swift-B network evaluation
ATT_FLOP_PER_EVAL = 12 * 8.858e9  # QK^T + PV of the 12 layers
PEAK_BF16, PEAK_F32 = 2.5e15, 157.3e12  # MI355X_MICROARCH.md: dense MFMA peaks
MAX_UNITS = 256  # include/swiftk.h: SWIFTK_MAX_UNITS


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="(member, IC) units per GPU per step (default 96 = configs[3]'s per-GPU share; 24 in --rollout mode)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x3"],
                    help="bf16 = BASELINE configs[1]'s dtype (default); f32 = exact-fp32 engine; bf16x3 = fp32-grade split-bf16 engine")
    ap.add_argument("--cpu-steps", type=int, default=3, help="samples of the CPU baseline, median reported (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip parity_engine / bf16_vs_fp32 / cpu_baseline legs")
    ap.add_argument("--graph", action="store_true",
                    help="replay the WHOLE step (noise draw included) as one HIP graph (pays off below ~8 units per step)")
    ap.add_argument("--solver", default="scm", choices=["scm", "2s", "dpm"],
                    help="scm = BASELINE configs[1] (default, the metric's workload); 2s / dpm = configs[2], multi-step ODE sampler")
    ap.add_argument("--num-steps", type=int, default=None, help="solver steps (default: 1 for scm, 20 for 2s, 8 for dpm)")
    ap.add_argument("--rollout", default=None, metavar="MEMBERSxICSxSTEPS",
                    help="north-star mode: roll MEMBERS x ICS units out to STEPS lead steps through RolloutEngine.run, units "
                         "sharded over the ranks (strong scaling); --steps/--warmup are ignored")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = 24 if a.rollout else 96
    if not 1 <= a.batch <= MAX_UNITS:
        ap.error(f"--batch must be in 1..{MAX_UNITS} (SWIFTK_MAX_UNITS: one swiftk_swinv2_forward call takes at most that many units)")
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    return a


def launch_ranks(n: int) -> int:
    """Bare `python bench.py --gpus N`: start N fresh rank processes (nothing in this process has touched the GPU) and
    return their worst exit code.  The reference's launch contract is one process per device (scripts/aurora-general.sh:74-91)."""
    import torch
    have = torch.cuda.device_count()  # counting devices does not initialise the GPU runtime
    if have < n and not os.environ.get("SWIFTK_ALLOW_SHARED_GPU"):  # (tests run N ranks on one GPU over gloo)
        print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    return wait_ranks(procs)


def wait_ranks(procs) -> int:
    """Worst exit code of the rank processes; when one fails the others (which would wait in a collective until the
    process-group timeout) are terminated -- by their own PIDs."""
    rc, live, stopped = 0, list(procs), set()
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if p.pid in stopped:  # (a sibling this launcher ended itself: its signal is not a result)
                continue
            rc = max(rc, abs(code))
            if code != 0:
                for q in live:
                    stopped.add(q.pid)
                    q.terminate()
    return rc


def build_net(dev, rank, grouped):
    import torch
    import torch.distributed as dist

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
    if grouped:  # weights travel once over RCCL/xGMI (north_star: broadcast for weights)
        for p in net.parameters():
            dist.broadcast(p.data, src=0)
    return net, state


def host_cpu():
    """(model name, physical cores, logical CPUs) of the host, from /proc/cpuinfo (unique (physical id, core id) pairs)."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
                phys = core = None
            elif k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            if phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return model, (len(cores) or logical), logical


def cpu_baseline(state, samples: int):
    """The oracle on the host cores: 1-member x 1-IC x 1-step forecasts (BASELINE config 1).  Thread counts {8, 32, physical
    cores} are swept with one timed sample each (oversubscribing a many-socket host with every logical CPU made the round-4
    figure slower than the reference on 8 threads); the fastest setting then runs `samples` in all and its median is reported."""
    import torch

    from oracle import sampler as osamp
    from oracle.swinv2 import OracleNet, SwinCfg
    from swift_amd.utils.detinit import det_normal

    cfg = SwinCfg(img_resolution=IMG, in_channels=2 * NV + NF, out_channels=NV, window_size=(16, 16), shift_size=(8, 8),
                  patch_size=(2, 2), depth=12, dim=1056, heads=12, auxiliary_dim=1)
    onet = OracleNet(cfg, state, NV, NV + NF)
    cond, lat = det_normal((1, NV + NF, *IMG), 1, "cond"), det_normal((1, NV, *IMG), 1, "lat")
    run = lambda: osamp.scm_solver(onet, lat, cond, 0.6, num_steps=1, sigma_min=0.02, sigma_max=200.0)
    model, physical, logical = host_cpu()
    before = torch.get_num_threads()
    sweep = sorted({min(n, logical) for n in (8, 32, physical)})

    def timed(n):
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        run()
        return time.perf_counter() - t0

    timed(sweep[-1])  # warm-up (first call pays allocator / oneDNN primitive creation)
    swept = {n: timed(n) for n in sweep}
    best = min(swept, key=swept.get)
    ts = [swept[best]] + [timed(best) for _ in range(max(samples - 1, 0))]
    torch.set_num_threads(before)
    med = sorted(ts)[len(ts) // 2]
    return dict(value=1.0 / med, unit="sample-steps/s", cores=best, cores_physical=physical, cpus_logical=logical, cpu_model=model,
                threads_swept={str(n): round(t, 2) for n, t in swept.items()}, kind="port",
                sample=f"median of {len(ts)} x (1 member x 1 IC x 1 step) on {best} threads, Swift-B scm 1-step, fp32, after 1 warm-up; "
                       f"{med:.2f} s per sample-step (all: {', '.join(f'{t:.2f}' for t in ts)}); seconds per sample-step by thread count in "
                       f"threads_swept; the reference itself measured 6.0 s on 8 threads of the build container (BASELINE.md section 2)")


def _cpu_topology():
    """[(logical cpu, socket, core)] from /proc/cpuinfo (empty when it does not say)."""
    out, cur = [], {}
    try:
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cur = {"cpu": int(v)}
            elif k == "physical id":
                cur["socket"] = int(v)
            elif k == "core id":
                cur["core"] = int(v)
                if "cpu" in cur and "socket" in cur:
                    out.append((cur["cpu"], cur["socket"], cur["core"]))
    except (OSError, ValueError):
        return []
    return out


def cpu_worker(threads: int, cpus: str, samples: int):
    """Child of cpu_baseline_parallel: one oracle forecast step at a time on its own CPUs; prints its per-step seconds as JSON."""
    if cpus:
        try:
            os.sched_setaffinity(0, {int(c) for c in cpus.split(",")})
        except OSError:
            pass
    import torch

    from oracle import sampler as osamp
    from oracle.swinv2 import OracleNet, SwinCfg
    from swift_amd.utils.detinit import det_normal, swinv2_state

    torch.set_num_threads(threads)
    state = swinv2_state(grid=(64, 128), in_channels=2 * NV + NF, out_channels=NV, patch_size=(2, 2), depth=12, dim=1056, heads=12, seed=1234)
    cfg = SwinCfg(img_resolution=IMG, in_channels=2 * NV + NF, out_channels=NV, window_size=(16, 16), shift_size=(8, 8),
                  patch_size=(2, 2), depth=12, dim=1056, heads=12, auxiliary_dim=1)
    onet = OracleNet(cfg, state, NV, NV + NF)
    cond, lat = det_normal((1, NV + NF, *IMG), 1, "cond"), det_normal((1, NV, *IMG), 1, "lat")
    run = lambda: osamp.scm_solver(onet, lat, cond, 0.6, num_steps=1, sigma_min=0.02, sigma_max=200.0)
    run()
    ts = []
    for _ in range(samples):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"seconds": ts}), flush=True)


def cpu_baseline_parallel(threads: int = 32, samples: int = 1, max_workers: int = 8):
    """The metric is a THROUGHPUT over independent (member, IC) units, so what the host can do is several oracle forecasts side by
    side, each on its own cores: K = physical cores / `threads` worker processes (whole workers per socket, first hardware thread
    of each core), every one running `samples` timed sample-steps after a warm-up, all started together.  Aggregate rate =
    sum over workers of 1 / (its median seconds per step) -- the workers overlap for all but their start-up skew."""
    topo = _cpu_topology()
    sockets = {}
    for cpu, sk, core in topo:
        sockets.setdefault(sk, {}).setdefault(core, cpu)  # first hardware thread of each physical core
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = None
    sets = []
    for sk in sorted(sockets):
        cpus = sorted(c for c in sockets[sk].values() if allowed is None or c in allowed)
        for k in range(len(cpus) // threads):
            sets.append(cpus[k * threads:(k + 1) * threads])
    sets = sets[:max_workers]
    if len(sets) < 2:
        return None
    me = os.path.abspath(__file__)
    procs = [subprocess.Popen([sys.executable, me, "--cpu-worker", str(threads), ",".join(map(str, cs)), str(samples)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                              env=dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))) for cs in sets]
    per = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
            per.append(json.loads(next(ln for ln in out.splitlines() if ln.startswith("{")))["seconds"])
        except Exception:  # noqa: BLE001 -- a worker that failed contributes nothing
            p.kill()
    if len(per) < 2:
        return None
    med = [sorted(t)[len(t) // 2] for t in per]
    return dict(value=sum(1.0 / m for m in med), unit="sample-steps/s", workers=len(per), threads_per_worker=threads, cores=len(per) * threads,
                seconds_per_step_by_worker=[round(m, 2) for m in med],
                what=f"{len(per)} oracle processes side by side, {threads} threads each on their own physical cores (whole workers per socket), "
                     f"{samples} timed sample-steps per worker after one warm-up; aggregate = sum of the workers' 1 / median")


def unit_inputs(units, dev, slabs: int = 1):
    """Initial standardised states [B, 69, H, W] and the pre-staged forcings [slabs, B, 3, H, W] (one slab
    per lead step, resident before the timed region); keyed by the unit's IC, so the members of an IC share them (as they
    share the dataset's files) and a unit's inputs do not depend on the rank that holds it."""
    import torch
    X = torch.empty(len(units), NV, *IMG, device=dev)
    F = torch.empty(slabs, len(units), NF, *IMG, device=dev)
    cache = {}
    for b, (ic, _m) in enumerate(units):
        if ic not in cache:
            g = torch.Generator(device=dev).manual_seed(1234 + ic)
            cache[ic] = (torch.randn(NV, *IMG, generator=g, device=dev), torch.randn(slabs, NF, *IMG, generator=g, device=dev))
        X[b] = cache[ic][0]
        F[:, b] = cache[ic][1]
    return X, F


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":  # child of cpu_baseline_parallel: never touches the GPU
        return cpu_worker(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]))
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))

    import torch
    import torch.distributed as dist

    from swift_amd import _lib, dist as sdist, ops
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import RolloutEngine, unit_seed

    # A process group is created for ONE rank too: the N = 1 run pushes the same collectives (weight broadcast, per-step
    # checksum all-gather, barriers, the max-over-ranks of the timing) through a one-rank RCCL communicator instead of
    # skipping them.  SWIFTK_SINGLE_RANK_GROUP=0 turns that off; an RCCL that cannot initialise is recorded, not fatal, at N = 1.
    rccl_error = None
    rccl_log = None
    if sdist.get_world_size() > 1 or os.environ.get("SWIFTK_RCCL_LOG"):
        # the first time N > 1 ranks meet, the record should show what RCCL built (ranks, channels, ring / tree orders)
        rccl_log = f"/tmp/swiftk_rccl.{os.getpid()}"
        os.environ["NCCL_DEBUG"] = "INFO"  # (forced: the pool's environment presets NCCL_DEBUG=VERSION)
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
        os.environ["NCCL_DEBUG_FILE"] = rccl_log + ".%h.%p.log"
    try:
        rank = sdist.setup_torch(single_rank_group=os.environ.get("SWIFTK_SINGLE_RANK_GROUP", "1") != "0")
    except Exception as e:  # noqa: BLE001
        if sdist.get_world_size() > 1:
            raise
        rank, rccl_error = 0, f"{type(e).__name__}: {e}"[:300]
    grouped = sdist.collectives_active()
    world = dist.get_world_size() if grouped else 1
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the process group has {world} rank(s) (WORLD_SIZE={os.environ.get('WORLD_SIZE')})",
              file=sys.stderr)
        sys.exit(3)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    dev = sdist.get_torch_device()
    lib = _lib.lib()
    # (kernel A/B knobs: SWIFTK_TUNE=key:value,... is applied by _lib.lib() when the library is loaded)
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "bf16x3": "bf16x3"}[a.dtype]
    B, K, W = a.batch, a.steps, a.warmup

    net, state = build_net(dev, rank, grouped)
    ds = SyntheticERA5Dataset([f"v{i}" for i in range(NV)], [f"f{i}" for i in range(NF)], img_resolution=IMG, length=64,
                              seed=1234)
    nsteps = a.num_steps or {"scm": 1, "2s": 20, "dpm": 8}[a.solver]
    evals = {"scm": nsteps, "2s": 2 * nsteps - 1, "dpm": nsteps}[a.solver]  # network evaluations per sample-step
    eng = RolloutEngine(net, ds, interval=6, solver=a.solver, denoise_dtype=dtype, num_steps=nsteps)
    rccl = {"world": world, "backend": dist.get_backend() if grouped else None,
            "version": (".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None) if grouped else None,
            "collectives": "weight broadcast, per-step all-gather of per-unit fp64 checksums, barriers, all-reduce(MAX) of the timing" if grouped else None}
    if rccl_error:
        rccl["error"] = rccl_error
    if rccl_log and rank == 0:
        rccl["topology_log"] = rccl_topology(rccl_log + ".*.log")

    def sync():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # what this process's HIP runtime does with a captured memset: recorded beside the numbers; the library itself
    # never captures one
    hip_rt = None
    if rank == 0 and not a.no_extras:
        try:  # (in a fresh child process: whether the stale slot is overwritten within a short probe depends on the process's state --
            # a fresh interpreter shows the HIP 7.0.x bug in a quarter of the replays, this one, with its pools warm, often in none)
            code = ("import json, sys; sys.path.insert(0, %r); from swift_amd.graphs import memset_node_probe; "
                    "print('PROBE ' + json.dumps(memset_node_probe()))" % os.path.dirname(os.path.abspath(__file__)))
            pr = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
            hip_rt = json.loads(next(ln for ln in pr.stdout.splitlines() if ln.startswith("PROBE "))[6:])
            hip_rt["what"] = ("a hipMemsetAsync captured into a HIP graph and replayed on the null stream with eager launches in between, in a fresh "
                              "process: memset_node_clean = False is the stale-fill-pattern bug of HIP 7.0.x that overflowed round 5's gradients "
                              "(tools/memset_graph_repro.hip, DESIGN section 11); kernel_clear_clean is the library's own clear under the same replay")
        except Exception as e:  # noqa: BLE001
            hip_rt = {"error": f"{type(e).__name__}: {e}"[:200]}
    if a.rollout:
        line = rollout_mode(a, eng, dev, rank, world, B, dtype, rccl, sync)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        return

    # this rank's units: a contiguous block of the flattened IC-major (IC, member) space
    units = [(u // MEMBERS, u % MEMBERS) for u in range(rank * B, rank * B + B)]
    n_slabs = W + K + 2  # one forcing slab per lead step of the run (warm-up, timed steps, the attention leg's two)
    X, forc_all = unit_inputs(units, dev, n_slabs)
    forc = forc_all[0].clone()  # the first slab, for the legs that roll a few units out on their own
    X0 = X.clone()
    mx, sx, st = eng.stats(dev)
    # latent noise: ONE launch per step for the whole batch, keyed by (unit seed, lead step, element) -- swiftk_unit_noise
    seeds = torch.tensor([unit_seed(m, ic) for ic, m in units], dtype=torch.int64, device=dev)
    lead = torch.zeros(1, dtype=torch.int64, device=dev)  # device-side lead-step counter (what a captured step reads)
    phys = torch.empty_like(X)
    z = torch.empty_like(X)
    ck = torch.zeros(B, dtype=torch.float64, device=dev)
    ck_all = torch.zeros(world * B, dtype=torch.float64, device=dev)
    ck_steps = torch.zeros(W + K + 4, dtype=torch.float64, device=dev)  # per step: sum over all units of the job
    step_no = [0]

    # per-rank split of the timed region (HIP events on the launch stream): step start -> in front of the output collective ->
    # behind it.  Summed per rank after the loop: compute_ms / collective_ms; barrier_wait_ms = the rank's idle time at the closing
    # barrier.  What a first multi-GPU run is read by; three event records per step.
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    timed = [False]

    def collect():
        """Output collection of one step: per-unit checksums of the physical state, gathered from every rank."""
        ops.unit_checksum(phys, out=ck)
        k = step_no[0] - W
        mark = timed[0] and 0 <= k < K
        if mark:
            ev[k][1].record()
        if grouped:
            dist.all_gather_into_tensor(ck_all, ck)
        else:
            ck_all.copy_(ck)
        if mark:
            ev[k][2].record()
        ck_steps[step_no[0]] = ck_all.sum()
        step_no[0] += 1

    def step():
        i = step_no[0]
        if timed[0] and 0 <= i - W < K:
            ev[i - W][0].record()
        ops.unit_noise(z, seeds, 0, step_dev=lead)
        ops.counter_add(lead, 1)
        Y = eng.sampler((X, forc_all[i % n_slabs]), latents=z)
        ops.rollout_update(X, Y, mx, sx, st, phys=phys)
        collect()

    if a.graph:  # the whole step is one replay: noise draw (device-side lead-step counter), network, state update
        fslab = forc_all[0].clone()  # static address inside the graph; the step's slab is copied in before the replay
        graph = eng.capture_step(X, fslab, z, phys, seeds=seeds, step=lead)

        def step():  # noqa: F811
            if timed[0] and 0 <= step_no[0] - W < K:
                ev[step_no[0] - W][0].record()
            fslab.copy_(forc_all[step_no[0] % n_slabs])
            graph.replay()
            collect()

    for _ in range(W):
        step()
    mlp2 = 2 * int(8 / 3.0 * 1056)
    if not a.graph:  # the per-launch event pairs of the roofline leg cannot be recorded inside a replayed graph
        lib.swiftk_profile_gemm(_lib.EPI_SWIGLU, mlp2)
    sync()
    timed[0] = True
    memset_prof = None
    if os.environ.get("SWIFTK_BENCH_MEMSET_SITES"):  # diagnosis: which host call sites issue device memsets inside a step?
        from torch.profiler import ProfilerActivity, profile
        memset_prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True)
        memset_prof.__enter__()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0  # this rank's own work is done; what follows is waiting for the slowest rank
    if memset_prof is not None:
        import collections
        memset_prof.__exit__(None, None, None)
        sites = collections.Counter()
        for e in memset_prof.events():
            if "memset" in e.name.lower() and e.device_type == torch.autograd.DeviceType.CPU:
                par, chain = e.cpu_parent, []
                while par is not None and len(chain) < 4:
                    chain.append(par.name)
                    par = par.cpu_parent
                sites[(e.name, " <- ".join(chain), " | ".join([s_ for s_ in (e.stack or []) if "site-packages" not in s_][:4]))] += 1
        print(f"MEMSET-SITES over {K} timed steps: {sum(sites.values())} runtime memset calls", file=sys.stderr)
        for (nm, chain, stack), c in sites.most_common(20):
            print(f"  x{c:4d} {nm} under [{chain}] at [{stack}]", file=sys.stderr)
    sync()
    dt = time.perf_counter() - t0
    timed[0] = False
    per_rank = sdist.gather_rank_times({
        "compute_ms": sum(e[0].elapsed_time(e[1]) for e in ev), "collective_ms": sum(e[1].elapsed_time(e[2]) for e in ev),
        "barrier_wait_ms": 1e3 * (dt - t_local), "timed_region_s": dt}, device=dev)
    tot_ms, n_launch = ctypes.c_double(0), ctypes.c_int64(0)
    lib.swiftk_profile_collect(ctypes.byref(tot_ms), ctypes.byref(n_launch))
    lib.swiftk_profile_gemm(-1, 0)
    if not torch.isfinite(phys).all():
        raise SystemExit("non-finite forecast state")
    ck_host = ck_all.cpu()
    ck_steps_host = ck_steps[W:W + K].cpu()
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
    if grouped:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    strong = None
    strong_spec = os.environ.get("SWIFTK_BENCH_STRONG", "12x64x60")  # (tests shrink the job; the key carries the spec)
    if world > 1 and (not a.no_extras or "SWIFTK_BENCH_STRONG" in os.environ) and a.solver == "scm" and nsteps == 1:
        # BASELINE configs[3] itself -- 12 members x 64 ICs x 60 steps, a FIXED job split over the ranks (strong scaling) -- beside
        # the weak-scaling headline, so that one driver invocation per N yields both curves
        import copy
        a3 = copy.copy(a)
        a3.rollout = strong_spec
        try:
            strong = rollout_mode(a3, eng, dev, rank, world, B, dtype, rccl, sync)
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline
            strong = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        # per-launch fabric traffic of the two roofline kernels from the committed PMC passes (valid for the profiled workload)
        traffic = {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pt = json.load(f)
            if pt.get("units_per_step") == B and a.dtype == "bf16":
                traffic = pt
        except OSError:
            pass
        value = world * B * K / dt
        M = B * 64 * 128
        flop_launch = 2.0 * M * mlp2 * 1056  # algorithmic: K = 1056, not the padded 1088
        avg_s = (tot_ms.value / max(n_launch.value, 1)) * 1e-3
        peak = PEAK_BF16 if a.dtype == "bf16" else PEAK_F32  # (bf16x3 is priced against the fp32 matrix peak it stands in for)
        ach = flop_launch / avg_s if avg_s > 0 else 0.0
        line = {
            "metric": "6h forecast steps/sec (members x ICs) on 128x256x69 ERA5",
            "value": value,
            "unit": "sample-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * dt / K,
            "timed_region_s": dt,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": ("Swift-B sCM 1-step sampler, 128x256x69 (BASELINE configs[1]): noise + fused network "
                                    "eval + residual state update + per-unit checksum collection per step") if a.solver == "scm" and nsteps == 1 else
                       f"Swift-B {a.solver} sampler, num_steps {nsteps} = {evals} network evaluations per sample-step, 128x256x69 "
                       "(BASELINE configs[2])", "units_per_gpu_per_step": B, "units": f"IC-major (IC, member) pairs, {MEMBERS} members per IC",
                       "hip_graph": bool(a.graph), "params": 225980976,
                       "noise": "swiftk_unit_noise: Philox4x32-10 + Box-Muller keyed by (member, IC, lead step), one launch per step",
                       "forcings": f"{n_slabs} pre-staged slabs [B, 3, 128, 256], one per lead step",
                       "parallelism": f"units sharded over {world} GPU(s), no data-path collective on the state"},
            "rccl": rccl,
            "hip_runtime": hip_rt,
            "per_rank": dict(per_rank, what="per rank, over the timed region: compute_ms = step start to the output collective (HIP "
                             "events on the launch stream), collective_ms = the all-gather of per-unit checksums, barrier_wait_ms = idle "
                             "at the closing barrier (the slowest rank shows ~0)"),
            "checksum": {"what": "fixed-order fp64 sum of each unit's physical state [69,128,256], all-gathered every step",
                         "units_collected": int(ck_host.numel()),
                         "last_step_rank0_units_sum": float(ck_host[:B].sum()),
                         "last_step_first_units": [float(v) for v in ck_host[:4]],
                         "last_step_all_units_sum": float(ck_host.sum()),
                         "per_step_all_units_sum": [float(v) for v in ck_steps_host]},
            "e2e": {"tflops": FLOP_PER_EVAL * evals * value / 1e12,
                    "frac_of_dense_mfma_peak": FLOP_PER_EVAL * evals * value / (peak * world)},
            "roofline": {"kernel": "gemm_kernel_p<bf16,bf16,SWIGLU> (w1 + SwiGLU)" if a.dtype == "bf16" else
                         "gemm_kernel_p<f32,f32,SWIGLU> (w1 + SwiGLU)", "bound": "mfma", "achieved": ach / 1e12,
                         "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic.get("gemm_swiglu"),
                         "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json)",
                         "launches": int(n_launch.value), "avg_launch_ms": avg_s * 1e3,
                         "flop_per_launch": flop_launch},
        }
        if strong is not None:  # (N > 1 only)
            line[f"rollout_{strong_spec}"] = ({k: strong[k] for k in ("value", "unit", "scaling", "timed_region_s", "per_rank", "config", "checksum")
                                              if k in strong} if "error" not in strong else strong)
        if att_n.value > 0:
            # The north star's named kernel.  Swift-B / bf16 runs to_qkv + cosine norm + shifted-window attention as ONE kernel
            # (swiftk_qkv_attention_fused: q, k, v never reach HBM), so it is judged against the MFMA roofline over the fused
            # FLOPs: 2 x 8192 x 3168 x 1056 (to_qkv) + 8.858e9 (QK^T + PV) per sample and layer.  Algorithmic bytes: the token
            # operand read once, the attention output written once, the weight once.  SWIFTK_TUNE=8:0 restores the two-kernel
            # path, whose attention kernel is HBM-bound (69.2 MB per sample-layer, 128 flop/B < ridge 312).
            att_s = att_ms.value / att_n.value * 1e-3
            fused = lib.swiftk_get_tuning(8) != 0  # read back from the library: the kernel that actually ran
            if fused:
                flop = B * (2.0 * 8192 * 3168 * 1056 + 8.858e9)
                bytes_ = B * 8192 * 2 * 1056 * 2.0 + 3168 * 1056 * 2.0
                line["attention_roofline"] = {
                    "kernel": "qkv_attn_kernel (to_qkv GEMM + cosine norm + shifted-window attention, one kernel per layer, bf16)",
                    "bound": "mfma", "achieved": flop / att_s / 1e12, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                    "frac": flop / att_s / PEAK_BF16, "traffic": traffic.get("qkv_attn_fused"), "launches": int(att_n.value),
                    "avg_launch_ms": att_s * 1e3, "flop_per_launch": flop, "algorithmic_bytes_per_launch": bytes_,
                    "attention_core_flop_per_launch": B * 8.858e9,
                    "note": "frac = fused FLOPs / time / the 2.5 PF dense peak (a 2.4 GHz figure).  SQ counters of this kernel at 96 units "
                            "(profiles/r05n_attn_counters_report.txt, round 5: ping-pong k-loop): MFMA pipe busy 0.60 of the SIMD cycles at an "
                            "effective clock of 1.86 GHz under this load (0.60 x 1.86 / 2.4 = 0.46; round 4: 0.53 at 1.92 GHz -- the chip gives "
                            "about half of a cycle saving back as clock); the attention core alone (fused minus a build with the core skipped, "
                            "0.73 ms of the 5.4) runs at 0.65 MFMA-busy with 4.5 VALU instructions per MFMA and no LDS bank conflicts.  Reading "
                            "head 0's weight slab in every item (L2-resident, wrong results) is 4.5 % faster: that is what the twelve slabs' "
                            "streaming from the Infinity Cache costs (profiles/r05i_qkv_attn_w0.txt; the head-grouped order that would keep slabs "
                            "in L2 measured +2 % time, profiles/r03e_*_ab8.txt).  Replaces swiftk_gemm_qkv_tiled + swiftk_window_attention, which "
                            "moved 10 GB of q/k/v per layer through HBM at 96 units",
                    "sq_counters": "profiles/r05n_attn_counters_report.txt (rocprofv3 --pmc at 96 units, bf16, default tuning; figures of that run, "
                                   "not of this one)"}
                if B == 96 and a.dtype == "bf16" and not os.environ.get("SWIFTK_TUNE"):  # the profiled configuration only
                    line["attention_roofline"]["mfma_pipe_busy"] = {
                        "kernel": 0.599, "attention_core": 0.648, "to_qkv_k_loop": 0.590, "plain_to_qkv_gemm": 0.522,
                        "effective_clock_ghz": 1.86, "source": "profiles/r05n_attn_counters_report.txt (rocprofv3 --pmc, 96 units)"}
            else:
                att_bytes, att_flop = B * 8192 * 4 * 1056 * 2.0, B * 8.858e9
                line["attention_roofline"] = {
                    "kernel": "attn_pipe_kernel (shifted-window attention, bf16, window-tiled q/k/v)", "bound": "hbm",
                    "achieved": att_bytes / att_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": att_bytes / att_s / 8e12,
                    "traffic": traffic.get("attention"), "launches": int(att_n.value), "avg_launch_ms": att_s * 1e3, "bytes_per_launch": att_bytes,
                    "mfma_tflops": att_flop / att_s / 1e12, "mfma_frac": att_flop / att_s / PEAK_BF16,
                    "note": "arithmetic intensity 128 flop/B < ridge 312: the HBM roofline caps MFMA utilisation at 41 % in bf16; the "
                            "MFMA-bound regime is the fp32 engine's kernel (parity_engine.attention_mfma_frac)"}
        if world == 1 and not a.no_extras and a.solver == "scm" and nsteps == 1:
            for key, leg in (("batch_sweep", lambda: batch_sweep_leg(eng.net, ds, dev, X0, forc, units, dtype)),
                             ("config3_2s", lambda: config3_leg(eng.net, ds, dev, X0, forc, units, dtype)),
                             ("rollout_12x8x60", lambda: rollout_leg(a, eng, dev, B, sync)),
                             ("variants", lambda: variants_leg(dev, dtype)),
                             ("parity_engine", lambda: parity_engine_leg(eng.net, ds, dev, lib, X0, forc, units)),
                             ("bf16_vs_fp32", lambda: drift_leg(eng.net, ds, dev, X0, forc, units))):
                try:
                    line[key] = leg()
                except Exception as e:  # noqa: BLE001 -- a reported extra must not take the metric's line down
                    line[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not a.no_extras and a.cpu_steps > 0:
            cb = cpu_baseline(state, a.cpu_steps)
            # the metric is a throughput over independent units: several oracle forecasts side by side use the host better than one
            # forecast on many threads (128 threads on one forecast are 2.8 x SLOWER than 32: NUMA).  The larger figure is the baseline.
            try:
                par = cpu_baseline_parallel(threads=int(cb["cores"]) if int(cb["cores"]) in (16, 32, 64) else 32)
            except Exception as e:  # noqa: BLE001
                par = {"error": f"{type(e).__name__}: {e}"[:200]}
            if par and "value" in par and par["value"] > cb["value"]:
                cb = dict(cb, single_process={k: cb[k] for k in ("value", "cores", "sample")}, value=par["value"], cores=par["cores"],
                          parallel=par, sample=par["what"] + "; one process alone: " + cb["sample"])
            elif par:
                cb["parallel"] = par
            line["cpu_baseline"] = cb
            line["vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        if world == 1 and not a.no_extras and a.solver == "scm" and nsteps == 1:
            # the children plan their resident activations around what other processes hold: hand the forecast path's
            # buffers (workspace for 96 units, states, engines) back first
            import gc
            del X, X0, forc, forc_all, phys, z, eng
            for e in list(getattr(net.model, "_engines", {}).values()):
                e._ws = None
            gc.collect()
            torch.cuda.empty_cache()
            try:
                line["generate_e2e"] = generate_e2e_leg()
            except Exception as e:  # noqa: BLE001
                line["generate_e2e"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            line["training"] = training_leg()
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def training_leg():
    """The training step that produces the forecast path's weights, measured by
    tools/train_bench.py in child processes (fresh allocator; they plan their resident activations around what this
    process still holds): one multistep-CRPS finetune iteration (BASELINE configs[4] per GPU), one sCM pre-training
    iteration and one TrigFlow iteration, Swift-B, local batch 8.  A reported extra, not the metric; a failure is recorded, not raised."""
    import subprocess

    import torch

    torch.cuda.empty_cache()
    out = {}
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "train_bench.py")
    # (six timed iterations each: with three, the first replays after the capture weighed 0.75-0.92 s on the CRPS figure)
    for name, args in (("crps_finetune_steps4", ["--loss", "crps", "--iters", "6"]), ("scm_pretrain", ["--loss", "scm", "--iters", "6"]),
                       ("trigflow", ["--loss", "trigflow", "--iters", "6"]),
                       ("scm_pretrain_muon", ["--loss", "scm", "--opt", "muon", "--iters", "6"])):  # the experiment's own optimiser
        try:
            p = subprocess.run([sys.executable, tool] + args, capture_output=True, text=True, timeout=600)
            rec = json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("{")))
            out[name] = {k: rec[k] for k in ("metric", "value", "unit", "samples_per_s", "what", "roofline", "peak_mem_gib", "allreduce") if k in rec}
        except Exception as e:  # noqa: BLE001 -- the forecast metric above must still be printed
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def generate_e2e_leg(members: int = 12, ics: int = 2, steps: int = 60):
    """The disk-included rate of the forecast job: ``python -m swift_amd.generate --synthetic`` as a child
    process -- the CLI a user runs (generate.py:23-43) -- for members x ics units x steps lead steps in ONE batch, once per raw
    output format, into a scratch directory that is removed afterwards.  The figure is the CLI's own clock around
    rollout_and_save (forcing staging, the device-resident rollout, device -> pinned ring -> store writes; model construction
    and imports excluded), so it sits beside the loop rate of this line's headline.  A reported extra; failures are recorded."""
    import re
    import shutil
    import subprocess
    import tempfile

    out = {"what": f"swift_amd.generate --synthetic --dtype bf16: {members} members x {ics} ICs x STEPS steps, one batch of {members * ics} "
                   "units, output streamed step by step through a pinned ring; rate = sample-steps / seconds inside rollout_and_save "
                   "(generate.py:48-154), store writes included", "unit": "sample-steps/s"}
    root = os.path.dirname(os.path.abspath(__file__))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    # the reference's job length (60 lead steps, generate.py:23-43): a 12-step job is over in a second and measures the CLI's start-up
    # (pinned ring, first launches: ~0.5 s) rather than its rate -- 24 units x 12 steps read 192-239 sample-steps/s where 36 steps read
    # 294-310 on the same box (profiles/r06zg_generate_e2e_job_length.txt).  The store is 14 GB per format at 60 steps: taken only
    # where the scratch filesystem has four times that free, else the short job
    need = 4 * (steps + 1) * members * ics * NV * IMG[0] * IMG[1] * 4
    try:
        if shutil.disk_usage(base or tempfile.gettempdir()).free < need:
            steps = 12
    except OSError:
        steps = 12
    out["what"] = out["what"].replace("STEPS", str(steps))
    for dump in ("numpy", "zarr"):
        d = tempfile.mkdtemp(prefix="swiftk_e2e_", dir=base)
        try:
            env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            p = subprocess.run([sys.executable, "-m", "swift_amd.generate", "--input", d, "--synthetic", "--members", str(members),
                                "--samples", str(ics), "--steps", str(steps), "--batch", str(members * ics), "--dtype", "bf16", "--dump", dump],
                               capture_output=True, text=True, timeout=600, env=env, cwd=root)
            m = re.search(r"Took ([0-9.]+) seconds: (\d+) sample-steps, ([0-9.]+) sample-steps/s", p.stdout + p.stderr)
            if p.returncode != 0 or not m:
                raise RuntimeError(f"rc {p.returncode}: {(p.stderr or p.stdout)[-200:]}")
            nbytes = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(d) for f in fs)
            out[dump] = {"value": float(m.group(3)), "seconds": float(m.group(1)), "sample_steps": int(m.group(2)),
                         "store_gb": nbytes / 1e9, "store_gb_per_s": nbytes / 1e9 / float(m.group(1)),
                         "scratch": "tmpfs" if base else "tmp"}
        except Exception as e:  # noqa: BLE001
            out[dump] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def _stepper(eng, dev, X0, forc, units, nb):
    """(step, X): `step()` advances the first nb units by one lead step (noise draw, network, state update) on `eng`."""
    import torch

    from swift_amd import ops
    from swift_amd.rollout import unit_seed

    mx, sx, st = eng.stats(dev)
    X, F = X0[:nb].clone(), forc[:nb].contiguous()
    seeds = torch.tensor([unit_seed(m, ic) for ic, m in units[:nb]], dtype=torch.int64, device=dev)
    z, phys = torch.empty_like(X), torch.empty_like(X)
    n = [0]

    def step():
        ops.unit_noise(z, seeds, n[0])
        n[0] += 1
        ops.rollout_update(X, eng.sampler((X, F), latents=z), mx, sx, st, phys=phys)

    return step, X


def _rate(step, nb, steps):
    import torch
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return nb * steps / (time.perf_counter() - t0)


def parity_engine_leg(net, ds, dev, lib, X0, forc, units, nb: int = 8, steps: int = 4):
    """The exact-fp32 engine (the configuration that meets the 1e-4 tolerance) on the first `nb` units of the workload, and
    the split-bf16 engine (`--dtype bf16x3`) beside it."""
    import torch

    from swift_amd import _lib
    from swift_amd.rollout import RolloutEngine

    nb = min(nb, X0.shape[0])
    eng = RolloutEngine(net, ds, interval=6, solver="scm", denoise_dtype=torch.float32, num_steps=1)
    step, Xe = _stepper(eng, dev, X0, forc, units, nb)
    step()
    lib.swiftk_profile_gemm(_lib.PROF_ATTENTION, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = ctypes.c_double(0), ctypes.c_int64(0)
    lib.swiftk_profile_collect(ctypes.byref(ms), ctypes.byref(n))
    lib.swiftk_profile_gemm(-1, 0)
    rate = nb * steps / dt
    att_s = ms.value / max(n.value, 1) * 1e-3
    split = None
    try:  # same units, same noise: distance after one step, then the rate
        eng3 = RolloutEngine(net, ds, interval=6, solver="scm", denoise_dtype="bf16x3", num_steps=1)
        s_e, x_e = _stepper(eng, dev, X0, forc, units, nb)
        s_3, x_3 = _stepper(eng3, dev, X0, forc, units, nb)
        s_e()
        s_3()
        rel = float((x_3.double() - x_e.double()).norm() / x_e.double().norm())
        r3 = _rate(s_3, nb, steps)
        split = {"value": r3, "unit": "sample-steps/s", "vs_exact_engine": r3 / rate,
                 "rel_l2_vs_exact_engine_after_1_step": rel,
                 "what": "--dtype bf16x3: the fp32 engine's kernels with every GEMM as bf16 MFMA products of (hi, lo)-split operands "
                         "(swiftk_split3); within 1e-4 rel-L2 of the reference golden on the Swift-B step (tests/test_gpu_model.py)"}
    except Exception as e:  # noqa: BLE001 -- an optional leg must not take the line down
        split = {"error": f"{type(e).__name__}: {e}"[:200]}
    return {"dtype": "f32", "what": "exact-fp32 MFMA engine (v_mfma_f32_16x16x4_f32 / 32x32x2_f32), 1e-4 parity configuration "
            "(tests/test_gpu_model.py: rel-L2 vs the reference golden and vs the reference run in fp64)", "units_per_step": nb,
            "steps": steps, "value": rate, "unit": "sample-steps/s", "tflops": FLOP_PER_EVAL * rate / 1e12,
            "frac_of_fp32_matrix_peak": FLOP_PER_EVAL * rate / PEAK_F32,
            "attention_kernel": "attn_f32_kernel<88>", "attention_avg_launch_ms": att_s * 1e3, "attention_launches": int(n.value),
            "attention_mfma_tflops": nb * 8.858e9 / att_s / 1e12 if att_s > 0 else None,
            "attention_mfma_frac": nb * 8.858e9 / att_s / PEAK_F32 if att_s > 0 else None,
            "split_bf16_engine": split}


def drift_leg(net, ds, dev, X0, forc, units, nb: int = 2, marks=(1, 10, 60)):
    """bf16 engine vs fp32 engine over an autoregressive rollout of the same units with the same noise."""
    import torch

    from swift_amd.rollout import RolloutEngine

    nb = min(nb, X0.shape[0])
    out = {}
    states = {}
    for name, dt_ in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        eng = RolloutEngine(net, ds, interval=6, solver="scm", denoise_dtype=dt_, num_steps=1)
        step, X = _stepper(eng, dev, X0, forc, units, nb)
        snaps = {}
        for i in range(1, max(marks) + 1):
            step()
            if i in marks:
                snaps[i] = X.clone()
        states[name] = snaps
    for i in marks:
        a32, a16 = states["f32"][i].double(), states["bf16"][i].double()
        out[f"rel_l2_after_{i}_steps"] = float((a16 - a32).norm() / a32.norm())
    out["what"] = (f"standardised state of {nb} units, bf16 engine vs exact-fp32 engine, same initial state, forcings and noise; "
                   "random-weight Swift-B (logit scales up to 100), one staged forcing slab")
    return out


def batch_sweep_leg(net, ds, dev, X0, forc, units, dtype, sizes=(1, 4, 8, 12, 16, 32), steps: int = 6):
    """the sCM 1-step sampler at {1, 4, 8, 12, 16, 32} units per step (12 = one initial condition's ensemble; eager launches and, for the
    launch-bound sizes, the whole step replayed as one HIP graph)."""
    import torch

    from swift_amd import ops
    from swift_amd.rollout import RolloutEngine, unit_seed

    eng = RolloutEngine(net, ds, interval=6, solver="scm", denoise_dtype=dtype, num_steps=1)
    out = {}
    for nb in sizes:
        if nb > X0.shape[0]:
            continue
        step, _ = _stepper(eng, dev, X0, forc, units, nb)
        rec = {"eager": _rate(step, nb, steps)}
        if nb <= 8:
            X, F = X0[:nb].clone(), forc[:nb].contiguous()
            z, phys = torch.empty_like(X), torch.empty_like(X)
            seeds = torch.tensor([unit_seed(m, ic) for ic, m in units[:nb]], dtype=torch.int64, device=dev)
            lead = torch.zeros(1, dtype=torch.int64, device=dev)
            g = eng.capture_step(X, F, z, phys, seeds=seeds, step=lead)
            rec["hip_graph"] = _rate(g.replay, nb, steps)
            del g
        rec["best"] = max(rec.values())
        rec["frac_of_dense_mfma_peak"] = FLOP_PER_EVAL * rec["best"] / PEAK_BF16
        out[str(nb)] = rec
    out["unit"] = "sample-steps/s"
    out["what"] = "Swift-B sCM 1-step sampler, units per step as keyed; whole step (noise, network, state update) per unit-step"
    return out


def variants_leg(dev, dtype, units: int = 32, steps: int = 4):
    """the reference's larger Swift variants (era5-swinv2-1.4-scm.yaml:29-36 -- dim 1280 / 1536, 16
    heads -> head_dim 80 / 96, depth 16) at FULL size on the same kernels: sCM 1-step sampler, `units` units per step, seeded
    random weights.  Both take the fused to_qkv + window attention kernel (qkv_attn_kernel<80> / <96>) since round 5."""
    import torch

    from swift_amd.generating.factory import sampler_factory
    from swift_amd.models.precond import PassPrecond
    from swift_amd.utils.detinit import swinv2_state

    def flops_per_eval(dim, depth):
        ntok, mlp = 64 * 128, int(8 / 3 * dim)
        layer = 2 * ntok * (dim * 3 * dim + dim * dim + dim * 2 * mlp + mlp * dim) + 4 * ntok * 256 * dim + 2 * 2 * dim * 2 * dim
        return 2 * ntok * 564 * dim + depth * layer + 2 * ntok * dim * 276

    out = {}
    for name, dim, heads, depth in (("468M", 1280, 16, 16), ("664M", 1536, 16, 16)):
        mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=depth,
                    dim=dim, heads=heads)
        net = PassPrecond(mcfg, img_resolution=list(IMG), img_channels=NV, condition_channels=NV + NF, auxiliary_dim=1)
        net.load_state_dict(swinv2_state(grid=(64, 128), in_channels=2 * NV + NF, out_channels=NV, patch_size=(2, 2), depth=depth,
                                         dim=dim, heads=heads, seed=7))
        net = net.to(dev).eval()
        sampler = sampler_factory("scm", net, denoise_dtype=dtype, num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
        g = torch.Generator(device=dev).manual_seed(0)
        cond = torch.randn(units, NV + NF, *IMG, generator=g, device=dev)
        with torch.no_grad():
            for _ in range(2):
                y = sampler(cond)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                y = sampler(cond)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        fl = flops_per_eval(dim, depth) * units
        out[name] = {"dim": dim, "heads": heads, "head_dim": dim // heads, "depth": depth, "units_per_step": units,
                     "sample_steps_per_s": units / dt, "ms_per_step": 1e3 * dt, "tflops": fl / dt / 1e12,
                     "frac_of_dense_mfma_peak": fl / dt / PEAK_BF16, "finite": bool(torch.isfinite(y).all()),
                     "fused_qkv_attention": True}
        del net, sampler, cond, y
        torch.cuda.empty_cache()
    out["what"] = ("the reference's commented larger variants at full size (depth 16), bf16 engine, sampler + network per unit-step "
                   "(no dataset / state update); parity: tests/test_gpu_model.py::test_forward_other_swift_variants_vs_oracle and "
                   "tests/test_gpu_kernels.py::test_fused_qkv_attention[80 / 96]")
    return out


def config3_leg(net, ds, dev, X0, forc, units, dtype, sizes=(1, 8)):
    """BASELINE configs[2]: the multi-step ODE sampler (dpm_solver_2s, num_steps 20 = 39 network evaluations per sample-step)."""
    import torch

    from swift_amd.rollout import RolloutEngine

    eng = RolloutEngine(net, ds, interval=6, solver="2s", denoise_dtype=dtype, num_steps=20)
    out = {}
    for nb in sizes:
        if nb > X0.shape[0]:
            continue
        step, _ = _stepper(eng, dev, X0, forc, units, nb)
        r = _rate(step, nb, 2)
        out[str(nb)] = {"value": r, "network_evals_per_s": 39 * r, "frac_of_dense_mfma_peak": FLOP_PER_EVAL * 39 * r / PEAK_BF16}
    out["unit"] = "sample-steps/s (one sample-step = 39 Swift-B evaluations)"
    out["what"] = "dpm_solver_2s, num_steps 20, sigma 0.02..200 (solver/2s.yaml), units per step as keyed, eager launches"
    return out


def rollout_leg(a, eng, dev, B, sync):
    """The per-GPU share of BASELINE configs[3] (12 members x 8 ICs x 60 six-hour steps = 5,760 sample-steps) through
    RolloutEngine.run, inside the driver-run line: the north-star loop itself (lead-step-indexed forcings, counter-based noise,
    state resident), where the headline loop above is its per-step body."""
    import copy
    a2 = copy.copy(a)
    a2.rollout = "12x8x60"
    r = rollout_mode(a2, eng, dev, 0, 1, B, None, None, sync)
    return {"value": r["value"], "unit": r["unit"], "timed_region_s": r["timed_region_s"], "workload": r["config"]["workload"],
            "batch": B, "frac_of_dense_mfma_peak": r["e2e"]["frac_of_dense_mfma_peak"], "checksum": r["checksum"]["all_units_sum"]}


def rccl_topology(log_glob: str, max_lines: int = 14):
    """What RCCL said about the communicator it built (NCCL_DEBUG=INFO, subsystems INIT + GRAPH, written to a file so that
    stdout keeps its one JSON line): rank count, channels, the ring / tree orders it chose.  Best effort: [] when nothing
    was logged."""
    import glob
    import re
    keep = []
    pat = re.compile(r"(nranks|nRanks|Channel|Ring|Tree|Trees|Connected|comm 0x|Using network|P2P|xGMI|XGMI|Algo|algo|Proto)")
    for path in sorted(glob.glob(log_glob))[:1]:
        try:
            with open(path, errors="replace") as f:
                for ln in f:
                    if pat.search(ln):
                        keep.append(re.sub(r"^\S+:\d+:\d+ \[\d+\] ", "", ln.strip())[:200])
                        if len(keep) >= max_lines:
                            break
        except OSError:
            pass
    return keep


def rollout_mode(a, eng, dev, rank, world, B, dtype, rccl, sync):
    """BASELINE configs[3] / north star: MEMBERS x ICS units rolled out STEPS lead steps through RolloutEngine.run, units
    sharded over the ranks in contiguous IC-major blocks (an IC's members share one rank and its forcings), batches of B
    units.  Per batch the ensemble statistics are reduced on the device and only their sums cross ranks."""
    import torch
    import torch.distributed as dist

    from swift_amd import dist as sdist, ops
    from swift_amd.rollout import unit_seed

    members, ics, steps = (int(v) for v in a.rollout.lower().split("x"))
    n_units = members * ics
    mine = [(u // members, u % members) for u in sdist.shard_units(n_units, rank, world)]
    forc_cache = {}

    def batch_inputs(units):
        g = {}
        X = torch.empty(len(units), NV, *IMG, device=dev)
        F = torch.empty(steps, len(units), NF, *IMG, device=dev)
        for b, (ic, _m) in enumerate(units):
            if ic not in g:
                gen = torch.Generator(device=dev).manual_seed(1234 + ic)
                g[ic] = (torch.randn(NV, *IMG, generator=gen, device=dev), torch.randn(steps, NF, *IMG, generator=gen, device=dev))
            X[b] = g[ic][0]
            F[:, b] = g[ic][1]
        return X, F

    batches = [mine[i:i + B] for i in range(0, len(mine), B)]
    # warm-up: one batch, two lead steps (kernel load, workspace, weights prepared)
    if batches:
        Xw, Fw = batch_inputs(batches[0])
        eng.run(Xw, Fw[:2].contiguous(), 2, seeds=[unit_seed(m, ic) for ic, m in batches[0]], keep_trajectory=False)
    staged = [batch_inputs(b) for b in batches[:1]]  # inputs resident before the timed region starts
    ck_sum = torch.zeros(1, dtype=torch.float64, device=dev)
    sync()
    t0 = time.perf_counter()
    for bi, units in enumerate(batches):
        X, F = staged[0] if bi == 0 else batch_inputs(units)
        final = eng.run(X, F, steps, seeds=[unit_seed(m, ic) for ic, m in units], keep_trajectory=False)
        ck_sum += ops.unit_checksum(final).sum()
    if sdist.collectives_active():
        gathered = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(gathered, ck_sum)
    else:
        gathered = ck_sum
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    per_rank = sdist.gather_rank_times({"own_work_s": t_local, "barrier_wait_ms": 1e3 * (dt - t_local), "units": float(len(mine))}, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if sdist.collectives_active():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    total = n_units * steps
    value = total / dt
    peak = PEAK_BF16 if a.dtype == "bf16" else PEAK_F32
    return {
        "metric": "6h forecast steps/sec (members x ICs) on 128x256x69 ERA5", "value": value, "unit": "sample-steps/s",
        "n_gpus": world, "steps": steps, "warmup": 0, "ms_per_step": 1e3 * dt / steps, "timed_region_s": dt,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"{members} members x {ics} ICs x {steps} six-hour steps, Swift-B sCM 1-step sampler, 128x256x69 "
                               "(BASELINE configs[3]) through RolloutEngine.run", "units": n_units, "units_per_rank": len(mine),
                   "batch": B, "sample_steps": total, "params": 225980976,
                   "parallelism": f"IC-major units sharded over {world} GPU(s); per-rank ensemble checksums all-gathered"},
        "rccl": rccl,
        "per_rank": dict(per_rank, what="per rank: seconds until its own units were done, idle milliseconds at the closing barrier, units it rolled out"),
        "checksum": {"what": "sum over all units of the fixed-order fp64 checksum of the final physical state",
                     "all_units_sum": float(gathered.sum())},
        "e2e": {"tflops": FLOP_PER_EVAL * value / 1e12, "frac_of_dense_mfma_peak": FLOP_PER_EVAL * value / (peak * world)},
    }


if __name__ == "__main__":
    main()
