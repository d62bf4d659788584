"""``trainer.profile=true``: step markers and a per-phase timing record for the first iterations.

The reference wraps its loop in ``torch.profiler`` with ``schedule(wait=2, warmup=2, active=5)``, logs the operator table
and writes ``rank0_prof.json`` on rank 0 (training/trainer.py:155-177, 329-331, 389-396).  An operator table says little
about a loop whose arithmetic is hand-written kernels behind one C call per pass, so the same switch here
  * opens a roctx range around every phase of an iteration (data / forward / backward / all-reduce / optimiser) -- under
    ``rocprofv3 --marker-trace --kernel-trace`` the kernels of the 5 active steps fall into named ranges;
  * brackets the phases of the active steps with HIP events on the compute stream and, once the 9 steps are over, logs the
    mean per phase and writes them as Chrome trace events to ``rank0_prof.json`` (the reference's file name; opens in the
    same viewers).
Nothing synchronises the host inside the profiled steps; the events are read once, after the last active step.
"""
from __future__ import annotations

import contextlib
import json
import os
from typing import Dict, List, Tuple

import torch

from .. import dist

WAIT, WARMUP, ACTIVE = 2, 2, 5  # the reference's schedule


class StepProfiler:
    def __init__(self, path: str = "rank0_prof.json"):
        self.path = path
        self.step_num = 0
        self.tot_num_steps = WAIT + WARMUP + ACTIVE
        self._events: List[Tuple[int, str, torch.cuda.Event, torch.cuda.Event]] = []
        self.summary: Dict[str, float] = {}
        self.done = False

    @property
    def active(self) -> bool:
        return not self.done and WAIT + WARMUP <= self.step_num < self.tot_num_steps

    @contextlib.contextmanager
    def phase(self, name: str):
        """roctx range (always, until the schedule is over) + HIP events (active steps only) around one phase."""
        if self.done:
            yield
            return
        torch.cuda.nvtx.range_push(f"swift/{name}")  # torch's nvtx shim is roctx on ROCm (libroctx64)
        rec = self.active
        if rec:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        try:
            yield
        finally:
            if rec:
                e1.record()
                self._events.append((self.step_num, name, e0, e1))
            torch.cuda.nvtx.range_pop()

    def step(self) -> None:
        self.step_num += 1
        if not self.done and self.step_num >= self.tot_num_steps:
            self._finish()

    def _finish(self) -> None:
        self.done = True
        torch.cuda.synchronize()
        if not self._events:
            return
        origin = self._events[0][2]
        trace, per_phase = [], {}
        for step, name, e0, e1 in self._events:
            ms = e0.elapsed_time(e1)
            per_phase.setdefault(name, []).append(ms)
            trace.append({"name": name, "cat": "swift_amd", "ph": "X", "pid": dist.get_rank(), "tid": 0,
                          "ts": 1e3 * origin.elapsed_time(e0), "dur": 1e3 * ms, "args": {"step": step}})
        self.summary = {k: sum(v) / len(v) for k, v in per_phase.items()}
        total = sum(self.summary.values())
        rows = "\n".join(f"  {k:<12s} {v:9.3f} ms  {100 * v / max(total, 1e-9):5.1f} %" for k, v in self.summary.items())
        dist.log0(f"prof summary stats (mean GPU time per phase over {ACTIVE} iterations, HIP events):\n{rows}\n  total        {total:9.3f} ms")
        if dist.get_rank() == 0:
            with open(os.path.join(os.getcwd(), self.path), "w") as f:
                json.dump({"traceEvents": trace, "displayTimeUnit": "ms",
                           "otherData": {"schedule": {"wait": WAIT, "warmup": WARMUP, "active": ACTIVE},
                                         "mean_ms_per_phase": self.summary}}, f)
        self._events = []


class _NoProfiler:
    done = True

    @contextlib.contextmanager
    def phase(self, name: str):
        yield

    def step(self) -> None:
        pass
