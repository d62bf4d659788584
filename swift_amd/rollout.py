"""Device-resident autoregressive ensemble rollout (the hot loop of reference generate.py:85-131).

What changes relative to the reference loop, and why it is the same computation:
  * the state never leaves HBM: no per-step ``.cpu()`` (generate.py:129) -- the physical
    trajectory is written by the update kernel straight into a device buffer and copied out
    once per batch;
  * forcings are staged (standardised) on the device up front instead of an h5 read per sample
    per step (generate.py:101-112);
  * ``cat[X, forcings]`` and ``cat[x_t, condition]`` are never built: state, forcings and the
    noisy latent go to the patch-gather kernel as three sources;
  * unstandardise-x + unstandardise-t + add + re-standardise (generate.py:120-131) is one kernel;
  * the work unit is a (member, IC) pair, not a member: any contiguous block of the flattened
    member x IC space can run as one batch, which is what lets 12 members fill 8 GPUs.
Noise: a counter-based stream keyed by (unit seed = f(member, IC index), lead step, element) --
``swiftk_unit_noise``, Philox4x32-10 + Box-Muller, ONE launch per step for the whole batch -- so
results do not depend on how units are sharded or batched, and the draw can sit inside a captured
step (the reference seeds a torch generator per member and consumes it in batch order,
generate.py:83 -- that coupling is deliberately not reproduced; parity tests inject latents
explicitly, and ``sampler(X, generator=g)`` still draws exactly like factory.py:52-56).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

from . import graphs, ops
from .generating.factory import sampler_factory


def unit_seed(member: int, ic: int) -> int:
    """The 64-bit Philox key of a (member, IC) unit: member in the high word, the IC's dataset index in the low one --
    collision-free, independent of rank and batch."""
    return ((int(member) & 0x7FFFFFFF) << 32) | (int(ic) & 0xFFFFFFFF)


def update_stats(dataset, interval: int, device):
    """(mean_x, std_x, std_t) for ``ops.rollout_update``, shared by the generate rollout and the in-training validation
    rollout.  Both un-standardise the state and re-standardise the sum with the DEFAULT delta (SST zeroed both times,
    data/era5.py:135-149) but the residual with ``delta = interval`` (generate.py:120-131): zeroed too unless the interval is
    24.  So an SST channel is handed over as mean 0 / std 0 -- ``swiftk_rollout_update`` reads std 0 as "the standardised
    value of this channel is forced to zero" -- with residual-std 0 (interval != 24) or the real one (interval 24: physical
    output = residual * std_t, next standardised state = 0, as the reference).  std_t is None for a non-residual dataset
    (generate.py:132-136, validate.py:112-116: the output is the next standardised state, the trajectory gets
    ``unstandardize_x`` of it).  (The multistep CRPS loss uses ONE delta throughout and takes ``dataset.rollout_stats``
    as it is: with delta 24 its SST channel keeps the real std_x, loss.py:402-406.)"""
    residual = getattr(dataset, "residual", False)
    mx, sx, st = dataset.rollout_stats(int(interval) if residual else 6, device)
    if "sea_surface_temperature" in dataset.variables:
        i = list(dataset.variables).index("sea_surface_temperature")
        mx, sx = mx.clone(), sx.clone()
        mx[i], sx[i] = 0.0, 0.0
    return mx, sx, (st if residual else None)


class LazyForcings:
    """Indexable like the [steps, B, n_forc, H, W] tensor of ``RolloutEngine.stage_forcings``; slab i is staged by a pool
    thread (pinned when the target is a GPU) and copied to the device when it is asked for."""

    def __init__(self, engine, ic_indices, steps: int, device, pool):
        self.engine, self.steps, self.device = engine, int(steps), torch.device(device)
        self.uniq = sorted(set(int(j) for j in ic_indices))  # members of one IC share its forcings: read each file once
        self.pos = torch.tensor([self.uniq.index(int(j)) for j in ic_indices])
        self.futs = [pool.submit(self._stage, i) for i in range(self.steps)]  # in step order: slab 0 comes first

    def _stage(self, i: int) -> torch.Tensor:
        eng = self.engine
        f = torch.stack([eng.dataset.get_forcings(j + int(i * eng.interval // 6)) for j in self.uniq], 0)
        f = eng.dataset.standardize_x(f)[self.pos].contiguous()
        return f.pin_memory() if self.device.type == "cuda" else f

    def __len__(self) -> int:
        return self.steps

    def __getitem__(self, i: int) -> torch.Tensor:
        return self.futs[i].result().to(self.device, non_blocking=True)


class RolloutEngine:
    def __init__(self, net, dataset, interval: int = 6, solver: str = "scm", denoise_dtype: torch.dtype = torch.float32,
                 **solver_kwargs):
        self.net, self.dataset, self.interval = net, dataset, int(interval)
        kw = dict(num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=interval / 10.0)  # generate.py:255-260
        kw.update(solver_kwargs)
        # re-noising draws of the multi-step consistency sampler (diffusion.py:452-455: randn_like between network calls) come
        # from the same counter-based stream as the latents while ``run`` drives the sampler: draw k of lead step i is keyed
        # (unit seed, i + (k << 40)), so multi-step rollouts are as independent of sharding and batching as the 1-step one
        self._draw = None  # (seeds_dev, lead step, draw counter) while run() is inside a step with the device stream
        if "randn_like" not in kw:
            kw["randn_like"] = self._randn_like
        self.sampler = sampler_factory(solver, net, denoise_dtype=denoise_dtype, **kw)
        self.renoises = solver == "scm" and int(kw["num_steps"]) > 1  # (diffusion.py:452-455: draws between network calls)
        self.residual = getattr(dataset, "residual", False)  # generate.py:76
        self._stats = None

    def _randn_like(self, like: torch.Tensor) -> torch.Tensor:
        if self._draw is None:  # outside run(), or latents injected by the caller: the reference's draw
            return torch.randn_like(like)
        seeds_dev, step, k = self._draw
        self._draw = (seeds_dev, step, k + 1)
        return ops.unit_noise(torch.empty_like(like), seeds_dev, step + ((k + 1) << 40))

    def stats(self, device):
        """``update_stats`` of this engine's dataset and interval, cached per device."""
        if self._stats is None or self._stats[0].device != device:
            self._stats = update_stats(self.dataset, self.interval, device)
        return self._stats

    def stage_forcings(self, ic_indices: Sequence[int], steps: int, device) -> torch.Tensor:
        """Standardised forcings [steps, B, n_forc, H, W] on the device (file index j + i*interval//6)."""
        rows = []
        uniq = sorted(set(int(j) for j in ic_indices))  # members of one IC share its forcings: read each file once
        pos = torch.tensor([uniq.index(int(j)) for j in ic_indices])
        for i in range(steps):
            f = torch.stack([self.dataset.get_forcings(j + int(i * self.interval // 6)) for j in uniq], 0)
            rows.append(self.dataset.standardize_x(f))
        return torch.stack(rows, 0)[:, pos].contiguous().to(device, non_blocking=True)

    def stage_forcings_lazily(self, ic_indices: Sequence[int], steps: int, device, pool) -> "LazyForcings":
        """The same slabs as ``stage_forcings``, produced step by step on ``pool`` (a thread pool) and handed over one lead step
        at a time: ``run`` indexes ``forcings[i]`` when it reaches step i, so a rollout starts as soon as step 0's forcings are
        staged instead of after all ``steps`` of them (60 file reads per IC for the 15-day job)."""
        return LazyForcings(self, ic_indices, steps, device, pool)

    @torch.no_grad()
    def capture_step(self, X: torch.Tensor, forc: torch.Tensor, z: torch.Tensor, phys: torch.Tensor,
                     seeds: Optional[torch.Tensor] = None, step: Optional[torch.Tensor] = None) -> "torch.cuda.CUDAGraph":
        """One forecast step (sampler + residual state update) recorded as a HIP graph over the caller's static tensors:
        replaying it advances ``X`` in place from the noise in ``z`` and the forcings in ``forc``.  With ``seeds`` (device
        int64 [B]) and ``step`` (device int64 scalar) the latent draw is part of the graph too: every replay fills ``z`` for
        lead step ``*step`` (``swiftk_unit_noise``) and advances the counter, so the WHOLE step is one replay; without them
        the caller fills ``z`` before replaying.  For the launch-bound regime (a few units per step: ~100 launches per
        network evaluation against ~5 ms of kernels); at 8+ units per step the launch queue already runs ahead of the kernels."""
        mx, sx, st = self.stats(X.device)
        draw = seeds is not None
        if draw and step is None:
            raise ValueError("capture_step(seeds=...) needs the device step counter as well")
        if draw and self.renoises:
            # only the latent draw lives on the counter stream: a multi-step sampler's re-noising would come from the default
            # generator inside the graph (replay-order dependent, unlike run()'s (seed, step + (k << 40)) draws)
            raise ValueError("capture_step(seeds=...) supports one-step samplers only: a multi-step sampler re-noises between its "
                             "evaluations, and those draws are not on the device counter stream")

        def body():
            if draw:
                ops.unit_noise(z, seeds, 0, step_dev=step)
                ops.counter_add(step, 1)
            Y = self.sampler((X, forc), latents=z)
            ops.rollout_update(X, Y, mx, sx, st, phys=phys)

        side = torch.cuda.Stream(device=X.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside the capture: operand preparation, workspace allocation
            keep = X.clone()
            step0 = step.clone() if draw else None
            for _ in range(2):
                body()
            X.copy_(keep)
            if draw:
                step.copy_(step0)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with graphs.capture(graph):
            body()
        X.copy_(keep)
        if draw:
            step.copy_(step0)
        return graph

    @torch.no_grad()
    def run(self, X0: torch.Tensor, forcings: torch.Tensor, steps: int, *, seeds: Optional[Sequence[int]] = None,
            latents: Optional[Callable[[int], torch.Tensor]] = None, out: Optional[torch.Tensor] = None,
            keep_trajectory: bool = True, after_step: Optional[Callable[[int, torch.Tensor], None]] = None) -> torch.Tensor:
        """Roll ``steps`` lead steps from the standardised state X0 [B, n_vars, H, W] (device, fp32).

        ``forcings`` [steps, B, n_forc, H, W] standardised, on the device.  ``latents(i)`` overrides the
        noise of step i (tests); otherwise unit b's latent of lead step i is the counter-based draw keyed by
        (``seeds[b]``, i) -- one ``swiftk_unit_noise`` launch per step for the whole batch.
        Returns the physical trajectory as a [B, steps+1, n_vars, H, W] view of a step-major buffer
        ``out`` [steps+1, B, ...] (or the final physical state if ``keep_trajectory`` is False).
        ``after_step(j, out[j])`` is called on the host as soon as lead step j (0 = the initial state) has been ENQUEUED:
        the caller may queue a device-to-host copy of that slab behind it (output streaming, generate.py:129).
        """
        dev = X0.device
        B, nv, H, W = X0.shape
        mx, sx, st = self.stats(dev)
        X = X0.contiguous().float().clone()
        if keep_trajectory:
            if out is None:  # step-major so that every lead step is one contiguous [B, C, H, W] block
                out = torch.empty(steps + 1, B, nv, H, W, dtype=torch.float32, device=dev)
            out[0] = self.dataset.unstandardize_x(X.clone())
            if after_step is not None:
                after_step(0, out[0])
        seeds_dev = zbuf = None
        if latents is None:
            seeds = list(seeds) if seeds is not None else list(range(B))
            seeds_dev = torch.tensor([int(s) for s in seeds], dtype=torch.int64, device=dev)
            zbuf = torch.empty(B, nv, H, W, dtype=torch.float32, device=dev)
        phys = torch.empty_like(X)
        for i in range(steps):
            z = latents(i) if latents is not None else ops.unit_noise(zbuf, seeds_dev, i)
            self._draw = None if latents is not None else (seeds_dev, i, 0)
            Y = self.sampler((X, forcings[i]), latents=z)
            ops.rollout_update(X, Y, mx, sx, st, phys=out[i + 1] if keep_trajectory else phys)
            if keep_trajectory and after_step is not None:
                after_step(i + 1, out[i + 1])
        self._draw = None
        return out.transpose(0, 1) if keep_trajectory else phys
