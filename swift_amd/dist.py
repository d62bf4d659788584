"""Process bootstrap: one process per GPU over RCCL (replaces the reference's un-vendored ``ezpz``;
SURVEY.md section 2 row 20).  Rank discovery from the torchrun / MPI environment, no arithmetic."""
from __future__ import annotations

import logging
import os
from typing import Optional

import torch
import torch.distributed as dist

_log = logging.getLogger("swift_amd")


def _env_int(*names, default=0):
    for n in names:
        if n in os.environ:
            return int(os.environ[n])
    return default


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else _env_int("RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK")


def get_world_size() -> int:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return _env_int("WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE", default=1)


def get_local_rank() -> int:
    return _env_int("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "PALS_LOCAL_RANKID", default=0)


def get_torch_device(as_torch_device: bool = True):
    if torch.cuda.is_available():
        d = torch.device("cuda", get_local_rank() % max(torch.cuda.device_count(), 1))
    else:
        d = torch.device("cpu")
    return d if as_torch_device else d.type


def _parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_cpus_of_gpu(pci_bus_id: str, sysfs: str = "/sys") -> Optional[set]:
    """The CPUs of the NUMA node a GPU hangs off (``<sysfs>/bus/pci/devices/<id>/numa_node`` -> ``.../node/nodeN/cpulist``), or
    None where the kernel does not say (node -1, no sysfs entry)."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as f:
            return _parse_cpulist(f.read()) or None
    except (OSError, ValueError):
        return None


def bind_to_gpu_numa_node(local_rank: Optional[int] = None, local_world: Optional[int] = None, sysfs: str = "/sys") -> Optional[set]:
    """Pin this rank's host threads (forcing readers, pinned-ring writers, the launch thread) to the CPUs next to its GPU, so that
    the 3.4 GB/s per rank of pinned staging / output traffic stays on the GPU's own socket -- the reference leaves placement to
    the launcher (scripts/aurora-general.sh:74-91 binds ranks with ``--cpu-bind``).  When several local ranks share a node its
    CPUs are dealt out evenly.  Best effort: returns the CPU set applied, or None (no GPU, no sysfs information, affinity not
    permitted, ``SWIFTK_NUMA_BIND=0``).  Called by ``setup_torch`` after the rank has selected its device."""
    if os.environ.get("SWIFTK_NUMA_BIND", "1") in ("", "0") or not hasattr(os, "sched_setaffinity"):
        return None
    if local_rank is None:
        local_rank = get_local_rank()
    if local_world is None:
        local_world = _env_int("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", default=get_world_size())
    try:
        n = torch.cuda.device_count()
        if n == 0:
            return None
        ids = [_pci_bus_id(i) for i in range(n)]
    except Exception:  # noqa: BLE001 -- placement is an optimisation, never an error
        return None
    mine = numa_cpus_of_gpu(ids[local_rank % n], sysfs) if ids[local_rank % n] else None
    if not mine:
        return None
    # ranks whose GPUs share this node split its CPUs (in rank order) instead of piling onto all of them
    same = [r for r in range(max(local_world, 1)) if ids[r % n] and numa_cpus_of_gpu(ids[r % n], sysfs) == mine]
    allowed = sorted(mine & set(os.sched_getaffinity(0)))
    if len(same) > 1 and len(allowed) >= len(same) and local_rank in same:
        k = same.index(local_rank)
        per = len(allowed) // len(same)
        allowed = allowed[k * per:(k + 1) * per]
    if not allowed:
        return None
    try:
        os.sched_setaffinity(0, allowed)
    except OSError:
        return None
    return set(allowed)


def _pci_bus_id(i: int) -> Optional[str]:
    """``0000:c1:00.0`` of device i, from the driver's sysfs listing via the properties torch caches (no kernel launch)."""
    try:
        p = torch.cuda.get_device_properties(i)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001
        return None


def collectives_active() -> bool:
    """True when a process group exists: the collectives of the data path (weight broadcast, output all-gather, gradient
    all-reduce, barriers) then run through it -- also in a group of ONE rank (``setup_torch(single_rank_group=True)``),
    which is how the RCCL calls of the N-rank job are exercised on a single GPU."""
    return dist.is_available() and dist.is_initialized()


def setup_torch(backend: Optional[str] = None, timeout_s: int = 1800, single_rank_group: Optional[bool] = None) -> int:
    """``ezpz.setup_torch`` equivalent: init the default process group when launched with >1 rank.

    ``backend`` accepts the reference's config values ("ddp", system/ampere.yaml:3): on a GPU it
    means RCCL (``nccl`` in torch), on CPU ``gloo``.

    ``single_rank_group`` (default: env ``SWIFTK_SINGLE_RANK_GROUP``): also create the group when the job has one rank, so the
    same collectives run (through a one-rank RCCL communicator) instead of being skipped; rendezvous is an in-process
    ``HashStore`` unless the launcher provided MASTER_ADDR / MASTER_PORT.
    """
    world = get_world_size()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(get_local_rank() % torch.cuda.device_count())
        if world > 1:  # one rank per GPU on one node: host threads next to the rank's own GPU (round 6)
            cpus = bind_to_gpu_numa_node()
            if cpus:
                _log.info("rank %d: host threads bound to %d CPUs of GPU %d's NUMA node", get_rank(), len(cpus), get_local_rank())
    if single_rank_group is None:
        single_rank_group = os.environ.get("SWIFTK_SINGLE_RANK_GROUP", "0") not in ("", "0")
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        be = "nccl" if use_cuda else "gloo"
        if backend in ("gloo", "nccl"):
            be = backend
        be = os.environ.get("SWIFTK_DIST_BACKEND", be)  # tests: N ranks sharing one GPU run their collectives over gloo
        import datetime
        kw = dict(rank=_env_int("RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK"), world_size=world,
                  timeout=datetime.timedelta(seconds=timeout_s))
        if world == 1 and "MASTER_PORT" not in os.environ:
            kw["store"] = dist.HashStore()
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
        if be == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())  # eager communicator: init errors surface here
        try:
            with _stdout_to_stderr():  # RCCL prints a version banner on C stdout when its first communicator comes up
                dist.init_process_group(be, **kw)
                if be == "nccl":
                    dist.barrier()
                    torch.cuda.synchronize()
        except Exception:
            # a group that came up but cannot carry a barrier must not stay behind: collectives_active() would report it
            # and every later collective would run on a broken communicator
            if dist.is_initialized():
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001 -- the original error is the one to report
                    pass
            raise
    return get_rank()


class _stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block, and C stdio is flushed before it is restored: whatever a native
    library printf()s meanwhile (RCCL's start-up banner sits in C stdout's buffer until exit otherwise) ends up on stderr,
    so that a launcher which prints ONE machine-readable line on stdout (bench.py) really prints one."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        import sys
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def maybe_launch_ranks(n_gpus: Optional[int], module: str) -> None:
    """``python -m <module> ... --gpus N`` from a bare shell: start N rank processes (one per GPU, the reference's launch
    contract: scripts/aurora-general.sh:74-91) and exit with their worst code.  Returns immediately inside a rank (WORLD_SIZE
    set by this function, torchrun or MPI) or when ``n_gpus`` is None / 1.  Must run before anything touches the GPU: the
    children are fresh interpreters, and this process never initialises the device."""
    import socket
    import subprocess
    import sys
    if not n_gpus or n_gpus <= 1 or any(k in os.environ for k in ("WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE")):
        if n_gpus and get_world_size() not in (1, n_gpus):
            raise SystemExit(f"--gpus {n_gpus} but launched with {get_world_size()} rank(s)")
        return
    have = torch.cuda.device_count()  # counting devices does not initialise the GPU runtime
    if 0 < have < n_gpus and not os.environ.get("SWIFTK_ALLOW_SHARED_GPU"):
        raise SystemExit(f"--gpus {n_gpus} but this node exposes {have} GPU(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    run_id = os.environ.get("HYDRA_RUN_ID")
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if run_id is not None:
            env["HYDRA_RUN_ID"] = run_id
        procs.append(subprocess.Popen([sys.executable, "-m", module, *sys.argv[1:]], env=env))
    import time
    rc, live, stopped = 0, list(procs), set()
    while live:  # a failed rank would leave the others waiting in a collective: stop them (by their own PIDs)
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
    raise SystemExit(rc)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def run_on_rank0(fn, *args, **kwargs):
    """Rank-0-only filesystem work followed by a barrier (reference utils/helpers.py:5-8)."""
    out = fn(*args, **kwargs) if get_rank() == 0 else None
    barrier()
    return out


def log0(msg: str, *args):
    if get_rank() == 0:
        _log.info(msg, *args)
        print(msg, *args, flush=True)


def shard_units(n_units: int, rank: int, world: int):
    """Contiguous block partition of the flattened (member, IC) space: ranks get floor or ceil(n/world) units."""
    base, rem = divmod(n_units, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_rank_times(local: dict, device=None) -> dict:
    """Per-rank timing record of a measured region, for the first multi-GPU run to be read at a glance: every rank hands in
    the same keys (milliseconds or seconds as named by the caller), rank order is preserved, and every rank gets
    ``{key: [value of rank 0, value of rank 1, ...]}`` back (one all-gather of a float64 vector; a process without a group
    gets one-element lists).  bench.py / generate use it for ``compute_ms``, ``collective_ms``, ``barrier_wait_ms``."""
    import torch
    keys = sorted(local)
    if not collectives_active():
        return {k: [float(local[k])] for k in keys}
    import torch.distributed as tdist
    world = tdist.get_world_size()
    dev = device if device is not None else (get_torch_device() if tdist.get_backend() == "nccl" else torch.device("cpu"))
    mine = torch.tensor([float(local[k]) for k in keys], dtype=torch.float64, device=dev)
    out = torch.zeros(world * len(keys), dtype=torch.float64, device=dev)
    tdist.all_gather_into_tensor(out, mine)
    out = out.view(world, len(keys)).cpu()
    return {k: [float(out[r, j]) for r in range(world)] for j, k in enumerate(keys)}

