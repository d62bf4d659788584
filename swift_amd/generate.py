"""Ensemble autoregressive rollout driver -- ``python -m swift_amd.generate`` (mirrors reference
src/swift/generate.py: same flags, same run-directory layout, same npy output layout).

    python -m swift_amd.generate --input RUN --checkpoint NAME --members 12 --steps 60 --samples 64 --interval 6 --dump numpy

reads ``RUN/.hydra/config.yaml`` (the saved composed config, generate.py:161) and
``RUN/checkpoints/checkpoint-NNNNNN.pt`` (``state["ema"]``, generate.py:225-226) and writes
``RUN/output/<ckpt>/output-{n}i-{steps}s-{members}m-{interval}h.npy`` with shape
(samples, members, steps+1, C, H, W) float32 (utils/io.py:237-259).

MI355X-first differences: the flattened IC-major (IC, member) space is
sharded in contiguous blocks over ranks (the reference shards members only, generate.py:79-81,
which caps 12 members at 6x on 8 GPUs) -- an IC's members sit next to each other, so a batch
reads each IC's state and forcing files once; rank 0 loads the checkpoint and broadcasts the
weights over RCCL; each rank rolls its units on the device (``RolloutEngine``) and writes its
own slices of the shared store (npy memmap, or one zarr chunk file per unit and variable:
``utils/zarrlite.py``); a barrier closes the job.  Additive flags: ``--solver``, ``--num-steps``,
``--dtype``, ``--synthetic`` (random-init weights + synthetic fields when no run directory
exists), ``--metrics`` (ensemble RMSE / CRPS / spread-skill against the dataset's own fields,
reduced per rank on the device and all-gathered: eval/metrics.py:39-134 without the round trip
through the store) and ``--gpus N`` (start the N ranks from a bare ``python -m`` call).
"""
from __future__ import annotations

import argparse
import os
import time
from glob import glob

import numpy as np
import torch
import torch.distributed as tdist

from . import dist
from .config import Cfg, instantiate, load_saved
from .rollout import RolloutEngine, unit_seed

parser = argparse.ArgumentParser()
parser.add_argument("--input", type=str, required=True, help="Input directory")
parser.add_argument("--checkpoint", type=str, default=None, help="Checkpoint name (default: latest)")
parser.add_argument("--members", type=int, default=1, help="Number of ensemble members")
parser.add_argument("--steps", type=int, default=8, help="Number of prediction steps")
parser.add_argument("--batch", type=int, default=32, help="(member, IC) units per device batch")
parser.add_argument("--samples", type=int, default=-1, help="Number of samples use")
parser.add_argument("--interval", type=int, default=6, choices=[6, 12, 24], help="Interval in hours")
parser.add_argument("--dump", type=str, default=None, choices=["zarr", "numpy", "none"],
                    help="Output format (default: zarr, the reference's default, for a one-rank job; with more than one rank the "
                         "default is 'none' + --metrics -- the ensemble metrics, reduced on the devices, and no raw trajectories: one "
                         "rank streams 3.4 GB/s of fp32 fields (330 sample-steps/s x 9 MB + host page faults ~3 GB/s per process), "
                         "eight of them 27 GB/s into ONE filesystem.  Ask for zarr / numpy explicitly to get the raw store at any rank count)")
# additive
parser.add_argument("--solver", type=str, default="scm", choices=["scm", "2s", "dpm"])
parser.add_argument("--num-steps", type=int, default=1, help="solver steps per forecast step")
parser.add_argument("--dtype", type=str, default="f32", choices=["f32", "bf16", "bf16x3"],
                    help="compute engine: f32 = exact fp32 MFMA (the reference's arithmetic, factory.py:11); bf16x3 = fp32-grade "
                         "(<= 1e-4 of the reference) from split-bf16 MFMA products, about twice as fast; bf16 = throughput engine")
parser.add_argument("--synthetic", action="store_true", help="random-init weights + synthetic data (no run dir needed)")
parser.add_argument("--metrics", action="store_true", help="ensemble metrics vs the dataset's fields -> evaluation_metrics.json")
parser.add_argument("--gpus", type=int, default=None, help="start this many ranks (one per GPU) when not launched by torchrun/mpiexec")


def get_ckpt_num(fpath: str) -> int:
    """utils/helpers.py:11-14."""
    return int(fpath.split(".pt")[-2].split("-")[-1])


def create_empty_numpy(ofile, dataset_len, n_channels, img_resolution, members, steps):
    """utils/io.py:237-259."""
    np.lib.format.open_memmap(ofile, dtype=np.float32, mode="w+",
                              shape=(dataset_len, members, steps + 1, n_channels, *img_resolution))


def synthetic_cfg() -> Cfg:
    from .config import compose
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")
    return compose(here, "train", ["data=era5-synthetic-1.4"])


def select_indices(n_dataset: int, samples: int, steps: int, interval: int):
    if samples == -1:
        return list(range(n_dataset))
    return np.linspace(0, n_dataset - 1 - (steps * interval // 6), num=samples, dtype=int).tolist()  # generate.py:179-184


def create_empty_zarr(ofile, dataset, indices, members, steps, interval):
    """utils/io.py:161-235 through the stdlib writer (no zarr / xarray in this image)."""
    from .utils import zarrlite
    lat, lon = dataset.get_lat_lon()
    times = np.array([dataset.get_time(int(i)) for i in indices], dtype="datetime64[ns]")
    zarrlite.create_forecast_store(ofile, list(dataset.variables), times, lat, lon, members, steps, interval=interval, batch=1)


def unit_of(u: int, members: int):
    """Flattened unit id -> (member, position of its IC in ``indices``); IC-major, so an IC's members are adjacent."""
    return u % members, u // members


def rollout_and_save(engine: RolloutEngine, dataset, indices, members: int, steps: int, ofile: str, device, args):
    """generate.py:48-154 with (member, IC) units instead of members as the sharded work item."""
    dump = getattr(args, "dump", "numpy") or "zarr"
    if dump == "numpy":
        store = np.lib.format.open_memmap(ofile, mode="r+")  # shape / data offset of the .npy the launcher created
        store_fd = os.open(ofile, os.O_WRONLY)
        store_off, store_shape = int(store.offset), tuple(store.shape)   # [samples, members, steps + 1, nv, H, W] float32
        del store
    elif dump == "zarr":
        from .utils import zarrlite
        var_channels = zarrlite.variable_channels(list(dataset.variables))
    rank, world = dist.get_rank(), dist.get_world_size()
    n_ic = len(indices)
    want_metrics = bool(getattr(args, "metrics", False))
    batch = int(args.batch)
    if want_metrics:  # an IC's members must meet in one batch: whole ICs per rank and per batch
        batch = max(members, batch // members * members)
        ic_range = dist.shard_units(n_ic, rank, world)
        mine = range(ic_range.start * members, ic_range.stop * members)
    else:
        mine = dist.shard_units(members * n_ic, rank, world)  # unit u = ic * members + member
    nv = len(dataset.variables)
    interval = engine.interval
    done = 0
    metric_sums = {}  # IC position -> [steps, nv, 4] float64 (swiftk_ensemble_sums rows)
    # Output streaming (the reference copies every step to the host synchronously, generate.py:129).
    from concurrent.futures import ThreadPoolExecutor
    writer = ThreadPoolExecutor(max_workers=1)
    on_gpu = torch.device(device).type == "cuda"   # (the host-logic tests drive this loop with a CPU stand-in engine)
    copy_stream = torch.cuda.Stream(device=device) if on_gpu else None
    pending = None  # CPU stand-in path only: the writer's future for the previous batch
    # GPU path: the output leaves STEP BY STEP -- as soon as a lead step of the batch is enqueued its [B, nv, H, W] slab is copied
    # on the side stream into one of RING pinned slabs (RING x 0.87 GB at 96 units, instead of two whole pinned trajectories:
    # 2 x 52 GB for 60 steps) and written from there while the following steps compute; at the end of a batch only its last
    # slabs are still on their way, so a job of ONE batch (12 members x 8 ICs on a GPU) overlaps its output as well
    RING = 4
    ring, ring_busy, ring_pos = [None], [None] * RING, [0]

    t_host = [0.0, 0.0]  # seconds in input staging / output writes (host side)

    def flush(p):
        """CPU stand-in path: one batch's whole trajectory [steps+1, B, ...] (the GPU path writes step by step, flush_step)."""
        host, us = p
        t1 = time.time()
        h = host.numpy()
        if dump == "numpy":
            # buffer is step-major [steps+1, B, ...]: every (step, unit) slab is contiguous on both sides, so the store is filled
            # by positional writes from the loader threads (pwrite releases the GIL; assigning into a memory map of the file
            # page-faults its way through fresh pages at ~3 GB/s however many threads share the mapping -- less than one GPU
            # produces: 330 sample-steps/s x 9 MB)
            slab = int(np.prod(store_shape[3:])) * 4
            per_unit = store_shape[2] * slab

            def put(kmi):
                k, (m, ic) = kmi
                base = store_off + (ic * store_shape[1] + m) * per_unit
                for i in range(h.shape[0]):
                    os.pwrite(store_fd, memoryview(h[i, k]).cast("B"), base + i * slab)

            list(loaders.map(put, enumerate(us)))
        else:  # one chunk file per unit and variable: independent files, written by a few threads
            list(loaders.map(lambda kmi: zarrlite.write_unit(ofile, var_channels, kmi[1][1], kmi[1][0], h[:, kmi[0]]),
                             enumerate(us)))
        t_host[1] += time.time() - t1

    def flush_step(ev, host, us, j):
        ev.synchronize()
        t1 = time.time()
        h = host.numpy()  # [B, nv, H, W] of lead step j
        if dump == "numpy":
            slab = int(np.prod(store_shape[3:])) * 4
            per_unit = store_shape[2] * slab
            list(loaders.map(lambda kmi: os.pwrite(store_fd, memoryview(h[kmi[0]]).cast("B"),
                                                   store_off + (kmi[1][1] * store_shape[1] + kmi[1][0]) * per_unit + j * slab),
                             enumerate(us)))
        else:
            list(loaders.map(lambda kmi: zarrlite.write_unit_step(ofile, var_channels, kmi[1][1], kmi[1][0], j, h[kmi[0]]),
                             enumerate(us)))
        t_host[1] += time.time() - t1

    def stream_out(us):
        """after_step hook of RolloutEngine.run for one batch."""
        def hook(j, slab_dev):
            B_ = slab_dev.shape[0]
            if ring[0] is None or ring[0].shape[1] < B_:
                for f in ring_busy:
                    if f is not None:
                        f.result()
                ring[0] = torch.empty(RING, B_, *slab_dev.shape[1:], dtype=torch.float32, pin_memory=True)
            slot = ring_pos[0] % RING
            ring_pos[0] += 1
            if ring_busy[slot] is not None:
                ring_busy[slot].result()  # the writer is done with this slab (back-pressure when the store is slower than the GPU)
            host = ring[0][slot, :B_]
            ready, ev = torch.cuda.Event(), torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready)
                host.copy_(slab_dev, non_blocking=True)
                ev.record(copy_stream)
            ring_busy[slot] = writer.submit(flush_step, ev, host, us, j)
        return hook

    def stage(s):
        """Host-side inputs of the batch starting at unit s (pinned tensors; runs on the reader thread, one batch ahead)."""
        t1 = time.time()
        units = [unit_of(u, members) for u in range(s, min(s + batch, mine.stop))]
        ics = [indices[ic] for _, ic in units]
        state = getattr(dataset, "get_state", None)  # one read per IC; dataset[j] would also read j's training target
        x0 = {int(j): (dataset.standardize_x(state(int(j))) if state else dataset[int(j)][0][0][:nv]) for j in set(ics)}
        X0 = torch.stack([x0[int(j)] for j in ics], 0)
        # GPU path: forcings are staged step by step behind the rollout's back (the first batch starts after step 0's slab, not
        # after all `steps` of them); the CPU stand-in of the host-logic tests takes the whole tensor
        forc = engine.stage_forcings_lazily(ics, steps, device, stagers) if on_gpu else engine.stage_forcings(ics, steps, "cpu")
        truth = None
        if want_metrics:  # verifying fields of every lead step, physical units, one copy per IC of the batch
            uniq = sorted(set(ic for _, ic in units))
            get = state if state else (lambda j: dataset.unstandardize_x(dataset[j][0][0][:nv]))
            jobs = [int(indices[ic]) + (i + 1) * interval // 6 for ic in uniq for i in range(steps)]

            def load_truth():  # file reads / field synthesis in parallel (numpy and h5 release the GIL)
                fields = list(loaders.map(get, jobs))
                tt = torch.stack(fields, 0).view(len(uniq), steps, *fields[0].shape)
                return tt.pin_memory() if on_gpu else tt

            # needed only once the batch has been rolled out: on the GPU path it loads while the rollout runs
            truth = stagers.submit(load_truth) if on_gpu else load_truth()
        if on_gpu:
            X0 = X0.pin_memory()
        t_host[0] += time.time() - t1
        return units, X0, forc, truth

    def batch_metrics(units, dev_buf, truth):
        """Per IC of the batch: swiftk_ensemble_sums of its members against the verifying fields at every lead step."""
        from .eval.metrics import ensemble_sums
        lat = dataset.get_lat_lon()[0]
        uniq = sorted(set(ic for _, ic in units))
        if hasattr(truth, "result"):
            truth = truth.result()
        for k, ic in enumerate(uniq):
            slots = [b for b, (_, i) in enumerate(units) if i == ic]
            assert len(slots) == members and slots == list(range(slots[0], slots[0] + members))
            pred = dev_buf[1:, slots[0]:slots[0] + members].contiguous()       # [steps, members, nv, H, W]
            metric_sums[ic] = ensemble_sums(pred, truth[k].to(pred.device, non_blocking=True), lat).double().cpu()

    reader = ThreadPoolExecutor(max_workers=1)
    loaders = ThreadPoolExecutor(max_workers=8)
    stagers = ThreadPoolExecutor(max_workers=4)  # forcing slabs, step by step (their own pool: output writes must not queue behind them)
    starts = list(range(mine.start, mine.stop, batch))
    nxt = reader.submit(stage, starts[0]) if starts else None
    for bi, s in enumerate(starts):
        units, X0, forc, truth = nxt.result()
        nxt = reader.submit(stage, starts[bi + 1]) if bi + 1 < len(starts) else None
        X0 = X0.to(device, non_blocking=True)
        if not on_gpu:
            forc = forc.to(device, non_blocking=True)
        traj = engine.run(X0, forc, steps, seeds=[unit_seed(m, indices[ic]) for m, ic in units],  # [B, steps+1, ...] view
                          **({"after_step": stream_out(units)} if on_gpu and dump != "none" else {}))
        dev_buf = traj.transpose(0, 1)      # the contiguous step-major buffer behind it
        if want_metrics:
            batch_metrics(units, dev_buf, truth)
        if on_gpu:
            dev_buf.record_stream(copy_stream)  # its last slabs may still be on their way to the host
            done += len(units)
            dist.log0(f"rank 0: {done}/{len(mine)} units")
            continue
        if dump == "none":  # metrics only: nothing leaves the device but the reduced sums
            done += len(units)
            dist.log0(f"rank 0: {done}/{len(mine)} units")
            continue
        # CPU stand-in engine (host-logic tests): the whole step-major trajectory goes to the writer thread at once
        host = dev_buf.contiguous()
        if pending is not None:
            pending.result()                # one trajectory in flight on the writer thread
        pending = writer.submit(flush, (host, units))  # host-side write under the next batch's staging
        done += len(units)
        dist.log0(f"rank 0: {done}/{len(mine)} units")
    if pending is not None:
        pending.result()
    for f in ring_busy:
        if f is not None:
            f.result()
    writer.shutdown()
    reader.shutdown()
    loaders.shutdown()
    stagers.shutdown()
    if dump == "numpy":
        os.close(store_fd)
    dist.log0(f"host side: {t_host[0]:.2f} s staging inputs, {t_host[1]:.2f} s writing outputs")
    if want_metrics:
        return collect_metrics(metric_sums, n_ic, steps, nv, members, dataset, interval, device, os.path.dirname(ofile))
    return None


def collect_metrics(metric_sums, n_ic, steps, nv, members, dataset, interval, device, odir):
    """Output collection over RCCL: every rank contributes the ensemble sums of its ICs ([n_ic, steps, nv, 4], zeros
    elsewhere), one all-gather, rank 0 turns them into the reference's metric names (eval/metrics.py:39-134)."""
    import json
    local = torch.zeros(n_ic, steps, nv, 4, dtype=torch.float64)
    for ic, v in metric_sums.items():
        local[ic] = v
    if dist.collectives_active():
        on_gpu = torch.device(device).type == "cuda"
        buf = local.to(device) if on_gpu else local
        parts = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
        tdist.all_gather(parts, buf)
        total = torch.stack(parts, 0).sum(0).cpu()
    else:
        total = local
    if dist.get_rank() != 0:
        return None
    H, W = dataset.img_resolution
    hw, N = H * W, members
    res = {}
    for j in range(steps):
        s = total[:, j]                                                   # [n_ic, nv, 4]
        rmse = torch.sqrt(s[..., 0] / hw).mean(0)
        crps = s[..., 1].sum(0) / (n_ic * N * hw) - (s[..., 2] / hw / (2 * N * (N - 1))).mean(0)
        ssr = torch.sqrt(s[..., 3] / hw).mean(0) / rmse
        for i, v in enumerate(dataset.variables):
            tag = f"{(j + 1) * interval}h"
            res[f"rmse_{v}_{tag}"], res[f"crps_{v}_{tag}"], res[f"ssr_{v}_{tag}"] = float(rmse[i]), float(crps[i]), float(ssr[i])
    path = os.path.join(odir, "evaluation_metrics.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    dist.log0(f"ensemble metrics of {n_ic} ICs x {members} members gathered from {dist.get_world_size()} rank(s): {path}")
    return res


def main(args):
    if args.synthetic and not os.path.exists(os.path.join(args.input, ".hydra", "config.yaml")):
        cfg = synthetic_cfg()
    else:
        cfg = load_saved(os.path.join(args.input, ".hydra", "config.yaml"))
    dist.setup_torch(backend=cfg.system.torch.backend)
    np.random.seed(cfg.seed % (1 << 31))
    torch.manual_seed(np.random.randint(1 << 31))
    device = dist.get_torch_device()

    dist.log0("Loading dataset...")
    dataset = instantiate(cfg.data.dataset, split="test", _convert_="object")
    indices = select_indices(len(dataset), args.samples, args.steps, args.interval)

    dist.log0("Constructing network...")
    net = instantiate(cfg.precond, model_config=cfg.model, img_resolution=dataset.img_resolution,
                      img_channels=dataset.n_target_channels, condition_channels=dataset.n_condition_channels,
                      sigma_max=float("inf"), _recursive_=False, _convert_="object")
    ckpt_basename = "synthetic"
    if not args.synthetic:
        if args.checkpoint is not None:
            name = args.checkpoint if args.checkpoint.endswith(".pt") else args.checkpoint + ".pt"
            ckpt = os.path.join(args.input, "checkpoints", name)
            if not os.path.exists(ckpt):
                raise ValueError(f"Specified checkpoint {ckpt} does not exist")
            ckpt_basename = os.path.basename(name)[:-3]
        else:
            paths = sorted(glob(os.path.join(args.input, "checkpoints", "checkpoint*.pt")), key=get_ckpt_num)
            assert paths, FileNotFoundError(f"No checkpoints in {os.path.join(args.input, 'checkpoints')}")
            ckpt, ckpt_basename = paths[-1], "latest"
        if dist.get_rank() == 0:
            dist.log0(f"Loading checkpoint: {ckpt}")
            net.load_state_dict(torch.load(ckpt, map_location="cpu", weights_only=True)["ema"])
    elif dist.get_rank() == 0:
        from .utils.detinit import swinv2_state
        m = net.model
        net.load_state_dict(swinv2_state(grid=m.grid_size, in_channels=m.in_channels, out_channels=m.out_channels,
                                         patch_size=m.patch_size, depth=m.depth, dim=m.dim, heads=m.heads,
                                         auxiliary_dim=m.auxiliary_dim, logvar=m.logvar_embed is not None, seed=cfg.seed))
    net = net.to(device).eval()
    if dist.collectives_active():  # one-time weight broadcast over RCCL / xGMI
        for p in net.parameters():
            tdist.broadcast(p.data, src=0)

    if args.dump is None:  # (see --dump: the raw store is the one-rank default, metrics only the multi-rank one)
        args.dump = "zarr" if dist.get_world_size() == 1 else "none"
        if args.dump == "none":
            args.metrics = True
            dist.log0(f"{dist.get_world_size()} ranks and no --dump given: writing the ensemble metrics only (--dump none --metrics); raw "
                      "trajectories stream at 3.4 GB/s per rank -- pass --dump zarr / numpy to get them")
    if args.dump == "none" and not args.metrics:
        raise ValueError("--dump none writes nothing but the metrics: add --metrics")
    odir = os.path.join(args.input, "output", ckpt_basename)
    dist.run_on_rank0(os.makedirs, odir, exist_ok=True)
    filename = f"output-{len(indices)}i-{args.steps}s-{args.members}m-{args.interval}h"
    if args.dump == "numpy":
        ofile = os.path.join(odir, f"{filename}.npy")
        dist.run_on_rank0(create_empty_numpy, ofile, len(indices), dataset.n_target_channels, dataset.img_resolution,
                          args.members, args.steps)
    elif args.dump == "none":
        ofile = os.path.join(odir, f"{filename}.none")  # (never created: names the directory evaluation_metrics.json goes to)
    else:  # zarr (the reference's default, generate.py:41-43)
        ofile = os.path.join(odir, f"{filename}.zarr")
        dist.run_on_rank0(create_empty_zarr, ofile, dataset, indices, args.members, args.steps, args.interval)

    solver_kwargs = dict(num_steps=args.num_steps, sigma_min=0.02, sigma_max=200.0, auxiliary=args.interval / 10.0)
    engine = RolloutEngine(net, dataset, interval=args.interval, solver=args.solver,
                           denoise_dtype={"bf16": torch.bfloat16, "f32": torch.float32, "bf16x3": "bf16x3"}[args.dtype],
                           **solver_kwargs)
    dist.log0("Rolling out samples...")
    t0 = time.time()
    rollout_and_save(engine, dataset, indices, args.members, args.steps, ofile, device, args)
    torch.cuda.synchronize()
    dist.barrier()
    el = time.time() - t0
    n = len(indices) * args.members * args.steps
    dist.log0(f"Done! Took {el:.3f} seconds: {n} sample-steps, {n / el:.1f} sample-steps/s including forcing staging and output "
              f"streaming to {os.path.basename(ofile)}.")
    if args.dump == "zarr" and dist.get_rank() == 0:  # generate.py:281-285
        from .utils import zarrlite
        zarrlite.consolidate(ofile)
    if tdist.is_initialized():
        tdist.destroy_process_group()
    dist.log0(f"Output saved to: {ofile if args.dump != 'none' else os.path.join(odir, 'evaluation_metrics.json')}")
    return ofile


if __name__ == "__main__":
    _args = parser.parse_args()
    dist.maybe_launch_ranks(_args.gpus, "swift_amd.generate")  # before anything touches the GPU; returns in the ranks
    main(_args)
