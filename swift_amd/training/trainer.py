"""Training loop (mirrors reference src/swift/training/trainer.py: same constructor kwargs, LR schedule, gradient
sanitising, EMA rule, tick bookkeeping and checkpoint format).

Data parallelism: one process per GPU; the fp32 gradients live in ONE flat buffer that every ``param.grad`` is a view of
and are averaged over RCCL / xGMI in per-layer slices of it: the explicit backward pass announces a layer's parameters as
soon as their gradients are final (``SwinTrainEngine.backward(grads_final=...)``, last backward pass of the iteration
only), ``GradAllReduce.reduce_params`` starts that slice's all-reduce asynchronously under the remaining layers, and
``sync()`` waits and reduces the rest (904 MB at Swift-B in 14 collectives).  The reference wraps the net in
``DistributedDataParallel(static_graph=True)`` (trainer.py:76-84), whose buckets do the same; ``GradAllReduce`` keeps its
``.module`` attribute and call signature so the losses' ``net.module`` accesses (loss.py:213) work unchanged.
"""
from __future__ import annotations

import copy
import json
import math
import os
import time
from typing import Iterable, Optional

import torch
import torch.distributed as tdist

from .. import dist, ops
from .loss import CRPSLoss, SCMLoss, TrigFlowLoss


class GradAllReduce(torch.nn.Module):
    """DDP stand-in: forwards to ``module``; ``sync()`` averages gradients across ranks in one collective."""

    def __init__(self, module: torch.nn.Module):
        super().__init__()
        self.module = module
        self._flat = None
        self._pending, self._ranges = [], []
        # what the first multi-GPU run needs to be read at a glance (per rank): how long the iteration's collectives would take
        # on their own (serial_ms, calibrate_serial), how long the compute stream actually waited for them in sync()
        # (exposed_ms), and how many bytes were announced early by the backward pass
        self._timing = []           # per sync(): (event before the waits, event behind them) or (t0, t1) host seconds on the CPU
        self._announced_elems = 0   # elements handed to reduce_params in the current iteration
        self._announced_hist = []
        self.serial_ms = None

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def flatten_grads(self):
        """Point every param.grad at a slice of one fp32 buffer (zero-filled)."""
        params = [p for p in self.module.parameters() if p.requires_grad]
        if self._flat is None:
            n = sum(p.numel() for p in params)
            dev = params[0].device
            # (the backward kernels ACCUMULATE into this buffer: on the GPU it is cleared by the library's own fill kernel)
            self._flat = ops.zeros_acc(n, device=dev) if dev.type == "cuda" else torch.zeros(n, dtype=torch.float32, device=dev)
            o = 0
            for p in params:
                p.grad = self._flat[o:o + p.numel()].view_as(p)
                o += p.numel()
        return self._flat

    def zero_grad_flat(self):
        flat = self.flatten_grads()
        if flat.is_cuda:
            ops.zero_acc_(flat)
        else:
            flat.zero_()
        self._pending, self._ranges = [], []

    @staticmethod
    def _world():
        return tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1

    @staticmethod
    def _active():
        """A process group exists (of any size: a one-rank RCCL group runs the same collectives, averaging over one rank)."""
        return tdist.is_available() and tdist.is_initialized()

    def _reduce(self, t, async_op=False):
        # RCCL averages in the collective; gloo (CPU tests) has no AVG: sum, then scale
        if tdist.get_backend() == "nccl":
            return tdist.all_reduce(t, op=tdist.ReduceOp.AVG, async_op=async_op), False
        return tdist.all_reduce(t, op=tdist.ReduceOp.SUM, async_op=async_op), True

    def reduce_params(self, params):
        """The gradients of ``params`` are final for this iteration: start their all-reduce now (one collective per run of
        neighbouring parameters in the flat buffer), overlapping the rest of the backward pass.  ``sync()`` waits for
        these and reduces whatever was never announced."""
        if not self._active() or self._flat is None or not params:
            return
        flat = self._flat
        spans = sorted(((p.grad.data_ptr() - flat.data_ptr()) // 4, p.numel()) for p in params)
        runs = []  # maximal runs of NEIGHBOURING announced parameters: what lies between two runs is not final yet
        for o, n in spans:
            if runs and runs[-1][1] == o:
                runs[-1][1] = o + n
            else:
                runs.append([o, o + n])
        for lo, hi in runs:
            if any(lo < b and a < hi for a, b in self._ranges):
                # the slice was averaged earlier in this iteration and a later backward pass has added rank-local gradients
                # on top of it: only the LAST backward pass of an iteration may announce parameters (gradient accumulation
                # and multi-term losses pass grads_final=None on all passes but the last; zero_grad_flat() starts an iteration)
                raise RuntimeError("GradAllReduce.reduce_params: parameters announced twice in one iteration "
                                   "(a backward pass ran after their gradients had been all-reduced)")
            handle, scale = self._reduce(flat[lo:hi], async_op=True)
            self._pending.append((handle, lo, hi, scale))
            self._ranges.append((lo, hi))
            self._announced_elems += hi - lo

    def _mark(self, flat):
        if flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time
        return time.perf_counter()

    def calibrate_serial(self, reps: int = 3, in_place: bool = False) -> float:
        """Milliseconds one blocking all-reduce of the whole flat gradient buffer takes (best of ``reps``): the yardstick the
        exposed wait of an iteration is compared with.  By default on a scratch buffer of the same size, so the gradients are
        not touched -- a second 4 B per parameter (2.7 GB at 664 M parameters), which a caller inside a running job must not
        allocate behind the memory plan's back (the CRPS loss sizes its resident rollout steps from free memory once):
        ``in_place=True`` reduces the gradient buffer itself, for callers that know its contents are dead (between an optimiser
        step and the next ``zero_grad_flat()``: the trainer's tick)."""
        flat = self.flatten_grads()
        if not self._active():
            self.serial_ms = 0.0
            return 0.0
        import time
        best = None
        # (a scratch copy of the same size: gloo's SUM would scale live gradients themselves)
        probe = flat if in_place else torch.zeros_like(flat)
        for _ in range(reps):
            if flat.is_cuda:
                torch.cuda.synchronize(flat.device)
            t0 = time.perf_counter()
            self._reduce(probe)
            if flat.is_cuda:
                torch.cuda.synchronize(flat.device)
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        del probe
        self.serial_ms = best
        return best

    def allreduce_stats(self) -> dict:
        """Per-rank record for the training JSON: bytes, the share the backward pass announced early, the serial time of the
        collective (None until calibrate_serial ran), the mean exposed wait of the recorded iterations and the overlap fraction
        1 - exposed / serial."""
        flat = self._flat
        if flat is None:
            return {}
        waits = []
        for a, b in self._timing:
            waits.append(a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3)
        exposed = sum(waits) / len(waits) if waits else None
        ann = self._announced_hist
        out = {"world": self._world(), "bytes": int(flat.numel()) * 4, "iterations_recorded": len(waits),
               "announced_early_frac": (sum(ann) / len(ann) / flat.numel()) if ann else 0.0,
               "serial_ms": self.serial_ms, "exposed_wait_ms": exposed,
               "overlap_frac": (max(0.0, min(1.0, 1.0 - exposed / self.serial_ms))
                                if exposed is not None and self.serial_ms else None)}
        return out

    def sync(self):
        flat = self.flatten_grads()
        world = self._world()
        if self._active():
            t_a = self._mark(flat)
            pending, self._pending = getattr(self, "_pending", []), []
            done = sorted(getattr(self, "_ranges", []))
            self._ranges = []
            pos = 0
            for a, b in done + [(flat.numel(), flat.numel())]:  # the complement of what the backward pass announced
                if a > pos:
                    _, scale = self._reduce(flat[pos:a])
                    if scale:
                        flat[pos:a].div_(world)
                pos = max(pos, b)
            for handle, lo, hi, scale in pending:
                handle.wait()
                if scale:
                    flat[lo:hi].div_(world)
            if len(self._timing) < 64:
                self._timing.append((t_a, self._mark(flat)))
                self._announced_hist.append(self._announced_elems)
            self._announced_elems = 0
        return flat


class Trainer:
    def __init__(
        self,
        net: torch.nn.Module,
        optimizer: torch.optim.Optimizer,
        loss_fn: torch.nn.Module,
        total_kimg: int = 200000,
        ema_halflife_kimg: int = 500,
        ema_rampup_ratio: Optional[float] = 0.05,
        lr_rampup_kimg: int = 10000,
        lr_min_factor: float = 0.01,
        lr_cosine_anneal: bool = True,
        kimg_per_tick: int = 50,
        checkpoint_ticks: Optional[int] = 50,
        device=None,
        amp_type: Optional[str] = "bfloat16",
        compile: bool = False,
        ckpt: Optional[str] = None,
        flop_count: Optional[int] = None,
        profile: bool = False,
        val_ticks=50,
        val_target_interval: int = 56,
        val_variables=None,
        net_pretrained=None,
        solver_kwargs: Optional[dict] = None,
        finetune_kwargs: Optional[dict] = None,
    ):
        self.device = torch.device(device) if device is not None else dist.get_torch_device()
        self.net = net.to(self.device)
        if amp_type not in ("bfloat16", None):
            raise NotImplementedError("the gfx950 training kernels take bf16 GEMM operands (amp_type: bfloat16)")
        # the reference's GradScaler is enabled for float16 only (trainer.py:72-75); with bf16 operands it is a disabled instance
        # whose state_dict() -- {} -- is what the checkpoint's "scaler" entry holds.  Kept as a real object so that saving and
        # resuming go through the same calls (a float16 run's scale found in a reference checkpoint is dropped, with a note).
        self.scaler = torch.GradScaler(self.device.type, enabled=False)
        self.ddp = GradAllReduce(self.net)
        self.ema = copy.deepcopy(net).eval().requires_grad_(False)
        self.base_lr = [g["lr"] for g in optimizer.param_groups]
        if ckpt is not None:  # trainer.py:104-116
            state = torch.load(ckpt, map_location=self.device, weights_only=True)
            self.net.load_state_dict(state["net"])
            self.ema.load_state_dict(state["ema"])
            self.resume_kimg = int(os.path.basename(ckpt).split("-")[1].split(".")[0])
            try:
                optimizer.load_state_dict(state["optimizer"])
            except ValueError:
                dist.log0("Could not load optimizer state, starting fresh.")
            if state.get("scaler"):  # trainer.py:109 -- a float16 run's loss scale has no meaning for bf16 operands
                dist.log0(f"checkpoint carries an enabled GradScaler state ({sorted(state['scaler'])}); bf16 training does not scale the loss")
            self.scaler.load_state_dict(state.get("scaler", {}))
        else:
            self.resume_kimg = 0
        self.loss_fn, self.optimizer = loss_fn, optimizer
        self.val_ticks, self.val_target_interval, self.val_variables = val_ticks, val_target_interval, val_variables
        self.solver_type, self.solver_kwargs = "dpm", dict(solver_kwargs or {})  # trainer.py:136-137
        self.net_pretrained = None if net_pretrained is None else net_pretrained.to(self.device).eval()  # trainer.py:119-123
        self.lr_rampup_kimg, self.lr_min_factor, self.lr_cosine_anneal = lr_rampup_kimg, lr_min_factor, lr_cosine_anneal
        self.finetune_kwargs = dict(finetune_kwargs or {})
        if self.finetune_kwargs.get("name") == "multistep":  # cumulative interval ends (trainer.py:140-145)
            cum = self.resume_kimg
            self.finetune_kwargs["intervals"] = [dict(iv) for iv in self.finetune_kwargs["intervals"]]
            for iv in self.finetune_kwargs["intervals"]:
                cum += iv["kimg"]
                iv["kimg"] = cum
        self.total_kimg, self.ema_halflife_kimg, self.ema_rampup_ratio = total_kimg, ema_halflife_kimg, ema_rampup_ratio
        self.kimg_per_tick, self.checkpoint_ticks = kimg_per_tick, checkpoint_ticks
        self.global_batch_size = None
        self._fused = None  # FusedAdamEMA once the flat gradient buffer exists (False: optimiser not covered)
        from .profiling import StepProfiler, _NoProfiler
        self.prof = StepProfiler() if profile else _NoProfiler()  # trainer.py:155-177

    # ------------------------------------------------------------------ one iteration
    def _get_batch(self, it):
        (x, t), (idx, delta) = next(it)
        # idx and delta stay on the host: the losses read them as numbers (forcing look-ups, rollout statistics) and move
        # what the network needs themselves -- a device copy would have to be read back, stalling the host behind the GPU
        return x.to(self.device, non_blocking=True), t.to(self.device, non_blocking=True), idx, delta

    def _forward_step(self, x, t, delta, **kwargs):
        with torch.autocast(self.device.type, enabled=True, dtype=torch.bfloat16):
            return self.loss_fn(self.ddp, t, condition=x, auxiliary=delta, **kwargs)

    def _set_lr(self, global_nimg: int):
        """linear warm-up from lr*min_factor, then cosine to lr*min_factor (trainer.py:201-217)."""
        warm = self.lr_rampup_kimg * 1000
        if global_nimg < warm:
            prog = global_nimg / warm
            for g, base in zip(self.optimizer.param_groups, self.base_lr):
                lo = base * self.lr_min_factor
                g["lr"] = lo + (base - lo) * prog
        elif self.lr_cosine_anneal:
            prog = min(1.0, (global_nimg - warm) / (self.total_kimg * 1000 - warm))
            for g, base in zip(self.optimizer.param_groups, self.base_lr):
                lo = base * self.lr_min_factor
                g["lr"] = lo + 0.5 * (base - lo) * (1 + math.cos(math.pi * prog))

    def _fused_optim(self, flat):
        """Adam / AdamW on device parameters: the whole step (sanitise, update, EMA) is one HIP kernel."""
        if self._fused is None:
            from . import fused_optim
            self._fused = False
            if flat.is_cuda and fused_optim.supported(self.optimizer):
                net_p = [p for p in self.net.parameters() if p.requires_grad]
                ema_of = {n: e for n, e in self.ema.named_parameters()}
                ema_p = [ema_of[n] for n, p in self.net.named_parameters() if p.requires_grad]
                if len(net_p) == len(list(self.net.parameters())):  # (a frozen parameter would still need its EMA lerp)
                    self._fused = fused_optim.FusedAdamEMA(self.optimizer, net_p, ema_p, flat)
        return self._fused

    def _backward_step(self, global_nimg: int, loss: torch.Tensor):
        self._set_lr(global_nimg)
        with self.prof.phase("backward"):
            loss.backward()
        with self.prof.phase("allreduce"):
            flat = self.ddp.sync()
        with self.prof.phase("optimizer"):
            self._optimizer_step(global_nimg, flat)

    def _optimizer_step(self, global_nimg: int, flat: torch.Tensor):
        half = self.ema_halflife_kimg * 1000
        if self.ema_rampup_ratio is not None:
            half = min(half, global_nimg * self.ema_rampup_ratio)
        beta = 0.5 ** (self.global_batch_size / max(half, 1e-8))
        fused = self._fused_optim(flat)
        if fused:
            fused.step(beta)  # trainer.py:223-246 in one pass over the parameters
            return
        # other optimisers (MuonWithAuxAdam, anything a config names): the reference's sequence on torch ops
        torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)  # trainer.py:223-231
        self.optimizer.step()
        with torch.no_grad():  # p_ema = p_net.lerp(p_ema, beta)  (trainer.py:245-246), all tensors in two multi-tensor launches
            pe, pn = list(self.ema.parameters()), [q.detach() for q in self.net.parameters()]
            torch._foreach_copy_(pe, torch._foreach_lerp(pn, pe, beta))

    def train_step(self, x, t, idx, delta, global_nimg: int, steps: int = 1):
        """One optimisation step on a prepared batch; returns the (rank-local) loss value."""
        self.ddp.zero_grad_flat()
        kw = {}
        if self.net_pretrained is not None:
            kw["net_pretrained"] = self.net_pretrained
        if isinstance(self.loss_fn, SCMLoss):      # trainer.py:382-386
            kw["step"] = global_nimg
        elif isinstance(self.loss_fn, CRPSLoss):
            kw.update(steps=steps, idx=idx)
        with self.prof.phase("forward"):
            loss = self._forward_step(x, t, delta, **kw)
        self._backward_step(global_nimg, loss)
        self.prof.step()  # trainer.py:389-396
        return loss.detach()

    # ------------------------------------------------------------------ validation (trainer.py:249-307)
    def _val_step(self, val_loader, val_dataset, cur_tick, global_nimg, val_stats_jsonl):
        from ..generating.factory import sampler_factory
        from .validate import RMSE_rollout
        kw = {k: v for k, v in self.solver_kwargs.items() if k != "intermediates"}
        sampler = sampler_factory(self.solver_type, self.ema, denoise_dtype=torch.bfloat16, **kw)
        agg, sep = RMSE_rollout(sampler, val_loader, val_dataset, self.val_target_interval, self.device, num_batches=1)
        agg_t = torch.tensor(agg, dtype=torch.float64, device=self.device)
        sep_t = torch.tensor(sep, dtype=torch.float64, device=self.device)
        if tdist.is_initialized():  # average across ranks
            tdist.all_reduce(agg_t)
            tdist.all_reduce(sep_t)
            agg_t /= dist.get_world_size()
            sep_t /= dist.get_world_size()
        rmse_map = dict(zip(val_dataset.variables, sep_t.cpu().numpy()))
        selected = [v for v in (self.val_variables or val_dataset.variables) if v in rmse_map] or list(val_dataset.variables)
        metrics = {"train/kimg": int(global_nimg / 1e3), "val/tick": cur_tick,
                   **{f"val/rmse/{v}": [float(x) for x in rmse_map[v]] for v in selected}, "val/rmse": float(agg_t)}
        dist.log0(json.dumps({k: metrics[k] for k in ("train/kimg", "val/tick", "val/rmse")}))
        if val_stats_jsonl is not None:
            val_stats_jsonl.write(json.dumps(metrics) + "\n")
            val_stats_jsonl.flush()
        return metrics

    # ------------------------------------------------------------------ loop
    def train(self, train_loader: Iterable, val_loader=None):
        it = iter(train_loader)
        val_dataset, val_stats = None, None
        if val_loader is not None:
            val_dataset = val_loader.sampler.dataset
            val_loader = iter(val_loader)
            if dist.get_rank() == 0:
                val_stats = open(os.path.join(os.getcwd(), "val_stats.jsonl"), "at")
        world = dist.get_world_size()
        global_nimg = self.resume_kimg * 1000
        tick_start_nimg, cur_tick, i = global_nimg, 0, 0
        t_start = tick_t0 = time.perf_counter()
        stats = open(os.path.join(os.getcwd(), "stats.jsonl"), "at") if dist.get_rank() == 0 else None
        steps = None
        while True:
            if self.finetune_kwargs.get("name") == "multistep":  # interval schedule (trainer.py:353-378)
                ivs = self.finetune_kwargs["intervals"]
                if steps is None or (global_nimg > ivs[0]["kimg"] * 1000 and len(ivs) > 1):
                    if steps is not None:
                        ivs.pop(0)
                    steps = ivs[0]["steps"]
                    sampler = getattr(getattr(train_loader, "batch_sampler", None), "sampler", None) or \
                        getattr(train_loader, "sampler", None)
                    if hasattr(sampler, "set_offset"):
                        sampler.set_offset(steps)
                        it = iter(train_loader)
            else:
                steps = 1
            with self.prof.phase("data"):
                x, t, idx, delta = self._get_batch(it)
            if self.global_batch_size is None:
                self.global_batch_size = x.shape[0] * world
            loss = self.train_step(x, t, idx, delta, global_nimg, steps)
            i += 1
            global_nimg += self.global_batch_size
            done = global_nimg >= self.total_kimg * 1000
            if not done and cur_tick != 0 and global_nimg < tick_start_nimg + self.kimg_per_tick * 1000:
                continue
            if self.val_ticks is not None and val_loader is not None and cur_tick % self.val_ticks == 0:  # trainer.py:411-418
                self._val_step(val_loader, val_dataset, cur_tick, global_nimg, val_stats)
            torch.cuda.synchronize()
            now = time.perf_counter()
            lv = loss.clone()
            if tdist.is_initialized():
                tdist.all_reduce(lv)
            metrics = {"train/tick": cur_tick, "train/iter": i, "train/loss": lv.item() / world,
                       "train/kimg": int(global_nimg / 1e3), "train/dt/dt": now - t_start, "train/dt/tick": now - tick_t0,
                       "train/dt/kimg": 1e3 * (now - tick_t0) / max(global_nimg - tick_start_nimg, 1),
                       "train/lr": self.optimizer.param_groups[0]["lr"]}
            if tdist.is_initialized():
                # per-rank gradient all-reduce record of this tick (what a first multi-GPU run is read by): serial time of the
                # collective (calibrated once, between iterations), mean exposed wait per iteration, overlap fraction
                # (in place: the optimiser step has consumed this iteration's gradients and the next iteration starts with
                # zero_grad_flat() -- no second gradient-sized buffer appears behind the CRPS loss's memory plan)
                if self.ddp.serial_ms is None and self.ddp._flat is not None:
                    self.ddp.calibrate_serial(in_place=True)
                mine = self.ddp.allreduce_stats()
                self.ddp._timing, self.ddp._announced_hist = [], []
                per_rank = [None] * world
                tdist.all_gather_object(per_rank, mine)
                metrics["train/allreduce"] = per_rank
            dist.log0(json.dumps(metrics))
            if stats is not None:
                stats.write(json.dumps(metrics) + "\n")
                stats.flush()
            if self.checkpoint_ticks is not None and (done or cur_tick % self.checkpoint_ticks == 0) and cur_tick != 0 \
                    and dist.get_rank() == 0:
                self._save_checkpoint(global_nimg)
            cur_tick += 1
            tick_start_nimg, tick_t0 = global_nimg, time.perf_counter()
            if done:
                if stats is not None:
                    stats.close()
                return metrics

    def _save_checkpoint(self, cur_nimg):
        """{"ema","net","optimizer","scaler"} -> checkpoints/checkpoint-{kimg:06d}.pt  (trainer.py:522-535)."""
        state = {"ema": self.ema.state_dict(), "net": self.net.state_dict(), "optimizer": self.optimizer.state_dict(),
                 "scaler": self.scaler.state_dict()}
        path = os.path.join(os.getcwd(), "checkpoints")
        os.makedirs(path, exist_ok=True)
        torch.save(state, os.path.join(path, f"checkpoint-{cur_nimg // 1000:06d}.pt"))
