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


def setup_torch(backend: Optional[str] = None, timeout_s: int = 1800) -> int:
    """``ezpz.setup_torch`` equivalent: init the default process group when launched with >1 rank.

    ``backend`` accepts the reference's config values ("ddp", system/ampere.yaml:3): on a GPU it
    means RCCL (``nccl`` in torch), on CPU ``gloo``.
    """
    world = get_world_size()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(get_local_rank() % torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        be = "nccl" if use_cuda else "gloo"
        if backend in ("gloo", "nccl"):
            be = backend
        import datetime
        dist.init_process_group(be, rank=_env_int("RANK"), world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return get_rank()


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
