"""HIP-graph replay of fixed launch sequences (training forward / backward / tangent passes).

A training iteration issues ~1000 kernels whose order, shapes and buffers never change; issued one by one from Python
(~100-150 us of interpreter and ctypes time per launch) the host needs about as long as the GPU (sCM pre-training at
Swift-B: 0.19 s of kernels per iteration).  ``GraphCache.call(key, fn, inputs)`` runs ``fn`` eagerly the first time a
key is seen (that run is the warm-up: its side effects -- gradient accumulation -- are real), captures it into a HIP
graph the second time, and from then on copies the inputs into the capture's static tensors and replays.  Everything
``fn`` allocates while being captured lives in a private memory pool, so the tensors it returned (activations)
stay valid until the next replay of the same key overwrites them: callers consume them before calling the key again,
which is the order the losses use (forward -> backward, one rollout step at a time).

Pools: a capture may place its tensors -- its live OUTPUTS included -- in blocks that an earlier capture of the same
pool used for temporaries, which is only safe when the graphs replay in capture order.  The losses do not guarantee that
(kept rollout slots are numbered from the end, so a longer rollout captures new slots late and replays them early), so
every key owns its pool -- except keys the caller declares ``transient``: their outputs are consumed before any other
sequence of this cache runs (backward passes: input gradients are copied or accumulated at once), so they share one pool
and with it their temporaries (several GB per backward pass at Swift-B).

Garbage collection: a cyclic-GC pass that happens to start INSIDE a capture can finalise objects of an earlier, invalidated
capture (graph executables, pool blocks), and the runtime aborts the process when such a release is issued on a capturing
stream (observed: SIGABRT in ``test_graph_pools_survive_growing_rollouts_and_moved_gradients`` once unrelated host code shifted
the collector's schedule).  ``capture()`` therefore keeps the collector off from capture begin to capture end; whatever became
garbage meanwhile is collected by the next ordinary pass, outside any capture.

Requirements on ``fn``: no host synchronisation, no Python-side dependence on tensor VALUES, every scalar kernel argument
constant for the key, and every buffer it reads besides ``inputs`` at a fixed address (the engines keep their operand
copies in persistent buffers for this).  ``SWIFTK_TRAIN_GRAPHS=0`` disables capture (everything runs eagerly).
"""
from __future__ import annotations

import contextlib
import gc
import os
from typing import Callable, Dict, Sequence

import torch


def enabled() -> bool:
    return os.environ.get("SWIFTK_TRAIN_GRAPHS", "1") != "0" and torch.cuda.is_available()


@contextlib.contextmanager
def capture(graph: "torch.cuda.CUDAGraph", pool=None):
    """``torch.cuda.graph`` with the cyclic garbage collector held off for the duration of the capture (see the module text)."""
    was_on = gc.isenabled()
    gc.disable()
    try:
        # thread_local: only THIS thread's calls are policed during the capture.  The default ("global") also invalidates it
        # when any other thread touches the runtime -- the DataLoader's pin-memory thread allocating a pinned batch, the
        # writer / reader threads of `generate` -- which is what a real training run does all the time
        # (hipErrorStreamCaptureInvalidated in the first captured pass of `python -m swift_amd.train` at Swift-B).
        with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
            yield
    finally:
        if was_on:
            gc.enable()


class GraphCache:
    def __init__(self):
        self._seen: Dict[tuple, int] = {}
        self._graphs: Dict[tuple, tuple] = {}
        self._transient_pool = None
        self.generation: Dict[tuple, int] = {}  # captures of a key so far: consumers of a key's output tensors (a backward
        #                                         sequence captured over a forward's activations) key themselves on it

    def invalidate(self) -> None:
        """Drop every captured sequence (a buffer they address was re-allocated); the next call of a key captures anew."""
        self._graphs.clear()
        self._transient_pool = None  # a pool dies with its last graph; a stale handle fails capture_begin

    def call(self, key: tuple, fn: Callable, inputs: Sequence[torch.Tensor], transient: bool = False):
        if not enabled() or torch.cuda.is_current_stream_capturing():
            return fn(*inputs)
        ent = self._graphs.get(key)
        if ent is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n == 0:
                return fn(*inputs)  # first sight: eager (doubles as the warm-up run)
            static_in = [x.clone() for x in inputs]
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if transient:
                if self._transient_pool is None:
                    self._transient_pool = torch.cuda.graph_pool_handle()
                pool = self._transient_pool
            else:
                pool = torch.cuda.graph_pool_handle()  # outputs outlive other keys' replays: a pool of its own
            with capture(graph, pool=pool):
                out = fn(*static_in)
            ent = self._graphs[key] = (graph, static_in, out)
            self.generation[key] = self.generation.get(key, 0) + 1
        graph, static_in, out = ent
        for s, x in zip(static_in, inputs):
            if s.data_ptr() != x.data_ptr():
                s.copy_(x)
        graph.replay()
        return out


def memset_node_probe(device=None, replays: int = 120) -> dict:
    """Does THIS process's HIP runtime replay a captured ``hipMemsetAsync`` correctly?  Under the runtime PyTorch 2.10.0+rocm7.0
    bundles (HIP 7.0.51831) a memset node replayed on the null stream -- torch's default stream -- writes whatever bytes later eager
    launches left where its fill pattern was (standalone: ``tools/memset_graph_repro.hip``); HIP 7.2 is clean.
    This library never captures a memset (``swiftk_zero_f32`` is a kernel), so the answer does not affect it: it is what a user
    who wraps these modules in graphs of their own wants to know, and what ``bench.py`` records beside its numbers.

    Two tiny graphs clear a 16-KB buffer -- one with the library's fill kernel, one with ``hipMemsetAsync`` (tuning key 25, restored
    afterwards) -- and are replayed in turn with eager launches in between whose scalar arguments are easy to recognise.  Returns
    {"kernel_clear_clean": bool (must be True), "memset_node_clean": bool, "memset_node_bad_replays": n, "replays": n}.  A clean
    answer for the memset node is weaker evidence than a dirty one: whether the stale slot is overwritten within the probe's few
    thousand launches depends on the process's state (a fresh process shows it in a quarter of the replays)."""
    from . import ops
    from ._lib import lib
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    L = lib()
    with torch.cuda.device(dev):
        ws_k, ws_m = torch.ones(4096, device=dev), torch.ones(4096, device=dev)
        a, b = torch.ones(1024, device=dev), torch.ones(1024, device=dev)
        host = torch.full((1024,), 3.0e38)
        before = L.swiftk_get_tuning(25)
        torch.cuda.synchronize()
        gk, gm = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        try:
            with capture(gk):
                ops.zero_acc_(ws_k)
            L.swiftk_set_tuning(25, 2)
            with capture(gm):
                ops.zero_acc_(ws_m)
        finally:
            L.swiftk_set_tuning(25, before)
        bad_k = bad_m = 0
        for r in range(replays):
            ws_k.fill_(1.0)
            ws_m.fill_(1.0)
            gk.replay()
            gm.replay()
            for k in range(96):  # eager traffic: its kernel arguments / staged copies are what a stale pattern picks up
                ops.axpby(3.0e38, a, -3.0e38, b, out=b)
                if k % 24 == 0:
                    a.copy_(host)  # (a pageable host-to-device copy: the runtime's staging path)
            bad_k += int(bool((ws_k != 0).any()))
            bad_m += int(bool((ws_m != 0).any()))
    return {"kernel_clear_clean": bad_k == 0, "memset_node_clean": bad_m == 0, "memset_node_bad_replays": bad_m, "replays": replays,
            "hip_runtime": getattr(torch.version, "hip", None)}
