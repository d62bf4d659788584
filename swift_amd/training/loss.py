"""TrigFlow and multistep-CRPS losses on the gfx950 training kernels (mirrors reference src/swift/training/loss.py).

Same constructor kwargs and call signatures as the reference's ``TrigFlowLoss`` (:117-160) and ``CRPSLoss`` (:306-445):
``loss = loss_fn(net_or_ddp, target, condition=x, auxiliary=delta, **kw)``; ``loss.backward()`` accumulates parameter
gradients.  The backward pass is not an autograd graph: the returned scalar hangs on a one-node ``autograd.Function``
whose backward runs the explicit kernels of ``train_engine`` (for the multistep loss: one rollout step at a time,
recomputing that step's activations first -- what the reference does with ``checkpoint_sequential``).

``SCMLoss`` (loss.py:163-260) gets its forward-mode tangent from ``jvp_engine.SwinJvpEngine`` (explicit tangent kernels,
csrc/jvp_kernels.hip) instead of ``torch.func.jvp``.
"""
from __future__ import annotations

import math
import os
from functools import partial
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .. import ops
from .._lib import check, lib
from ..models.precond import _process_auxiliary
from ..train_engine import SwinTrainEngine

PRESSURE_LEVELS = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
LEVEL_VARS = ["geopotential", "u_component_of_wind", "v_component_of_wind", "vertical_velocity", "wind_speed", "temperature",
              "relative_humidity", "specific_humidity", "vorticity", "potential_vorticity"]
SURFACE_W = {"2m_temperature": 1.0, "sea_surface_temperature": 0.1, "10m_u_component_of_wind": 0.1,
             "10m_v_component_of_wind": 0.1, "mean_sea_level_pressure": 0.1}


def _calculate_latitude_weights(lat_dim: int) -> torch.Tensor:
    """cos(latitude) / mean, floored at 0.1 -> [1,1,H,1]  (loss.py:28-32)."""
    w = torch.cos(torch.deg2rad(torch.linspace(-90, 90, lat_dim)))
    return torch.clamp(w / w.mean(), min=0.1).view(1, 1, -1, 1)


def _calculate_variable_weights(variables: Sequence[str]) -> torch.Tensor:
    """surface table + level/sum(levels) per pressure level, normalised to 1 -> [1,C,1,1]  (loss.py:35-55)."""
    tot = float(sum(PRESSURE_LEVELS))
    table = dict(SURFACE_W)
    for v in LEVEL_VARS:
        for lev in PRESSURE_LEVELS:
            table[f"{v}_{lev}"] = lev / tot
    w = torch.tensor([table.get(v, 1.0 / max(len(variables), 1)) for v in variables], dtype=torch.float32).view(1, -1, 1, 1)
    return w / w.sum()


def lognormal(x: torch.Tensor, P_mean: float, P_std: float) -> torch.Tensor:
    return torch.exp(torch.randn([x.shape[0], 1, 1, 1], device=x.device) * P_std + P_mean)


def loguniform(x: torch.Tensor, sigma_min: float, sigma_max: float) -> torch.Tensor:
    lo, hi = math.log(sigma_min), math.log(sigma_max)
    return torch.exp(lo + torch.rand([x.shape[0], 1, 1, 1], device=x.device) * (hi - lo))


NOISE_SAMPLING_METHODS = {"lognormal": lognormal, "loguniform": loguniform}


class _Deferred(torch.autograd.Function):
    """Scalar whose .backward() runs an explicit backward pass (accumulating into param.grad)."""

    @staticmethod
    def forward(ctx, anchor, runner):
        ctx.runner = runner
        return runner.value.detach().clone()

    @staticmethod
    def backward(ctx, g):
        ctx.runner.run_backward(g.detach().float())  # stays a device scalar: reading it would stall the host behind the GPU
        return None, None


def _engine(net) -> SwinTrainEngine:
    mod = getattr(net, "module", net)
    eng = getattr(mod.model, "_train_engine", None)
    if eng is None:
        eng = SwinTrainEngine(mod.model)
        mod.model._train_engine = eng
    return eng


def _memory_budget(dev, frac: float = 0.75) -> float:
    """Bytes this process may plan with: ``frac`` of the device, less what OTHER processes hold on it right now (a second
    rank sharing the GPU in a test, the parent of a benchmark child)."""
    free, total = torch.cuda.mem_get_info(dev)
    others = max(0, total - free - torch.cuda.memory_reserved(dev))
    return frac * total - others


def _plan_once(cache: dict, key, decide: Callable[[], int], what: str, dev) -> int:
    """A memory-planning decision (how many rollout steps keep their activations, one- or two-pass sCM) is taken ONCE per
    batch signature, at first sight, and reused: the instantaneous free-memory figure it starts from moves between
    iterations (RCCL buffers, the HIP context), and a decision that flips creates new capture keys mid-run.  Every rank
    must walk the same path (the per-layer all-reduces pair up across ranks), so with a process group the ranks agree on
    the minimum."""
    if key not in cache:
        v = int(decide())
        from ..dist import collectives_active
        if collectives_active():
            import torch.distributed as tdist
            tv = torch.tensor([v], dtype=torch.int64, device=dev if tdist.get_backend() == "nccl" else "cpu")
            tdist.all_reduce(tv, op=tdist.ReduceOp.MIN)
            v = int(tv.item())
        cache[key] = v
        import logging
        logging.getLogger("swift_amd").info("%s for %s: %d", what, key, v)
    return cache[key]


class _LossBase(torch.nn.Module):
    def __init__(self, dataset, sigma_data: float):
        super().__init__()
        self.dataset = dataset
        self.sigma_data = sigma_data
        self._plans = {}
        self.register_buffer("w_lat", _calculate_latitude_weights(dataset._shape[1]))
        self.register_buffer("w_var", _calculate_variable_weights(dataset.variables))

    def _w(self, dev):
        return self.w_var.reshape(-1).to(dev).contiguous(), self.w_lat.reshape(-1).to(dev).contiguous()

    @staticmethod
    def _anchor(dev):
        return torch.zeros((), device=dev, requires_grad=True)


class TrigFlowLoss(_LossBase):
    """loss = mean_{b,h,w} sum_c [ e^{-lv} w_var w_lat (sigma_d F - v_t)^2 + lv ]   (loss.py:132-160)."""

    def __init__(self, dataset, noise: dict, sigma_data: float):
        super().__init__(dataset, sigma_data)
        self.cfg = dict(noise)
        self._sampling_fn = partial(NOISE_SAMPLING_METHODS[self.cfg.pop("dist")], **self.cfg)

    def forward(self, net, x, condition=None, auxiliary=None, _tau=None, _z=None, **kwargs):
        mod = getattr(net, "module", net)
        eng = _engine(net)
        dev = x.device
        B, C, H, W = x.shape
        sd = float(self.sigma_data)
        tau = self._sampling_fn(x) if _tau is None else _tau
        t = torch.atan(tau.reshape(B).float() / sd).contiguous()
        z = (torch.randn_like(x) if _z is None else _z).contiguous().float()
        x = x.contiguous().float()
        xt, vt = torch.empty_like(x), torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        check(lib().swiftk_trigflow_prep(x.data_ptr(), z.data_ptr(), t.data_ptr(), xt.data_ptr(), vt.data_ptr(), sd, B, C * H * W,
                                         st), "swiftk_trigflow_prep")
        aux = _process_auxiliary(auxiliary, mod.auxiliary_dim, B, dev)
        want_lv = mod.model.logvar_embed is not None
        srcs, scales = [xt], [1.0]
        if condition is not None and mod.condition_channels > 0:
            srcs.append(condition)
            scales.append(1.0)
        res = eng.forward(srcs, scales, t, aux, want_logvar=want_lv)
        Fx, lv, ctx = res if want_lv else (res[0], None, res[1])
        wv, wl = self._w(dev)
        loss = ops.zeros_acc(1, device=dev)  # (atomically accumulated: cleared by the library's own kernel)
        dF = torch.empty_like(Fx)
        dlv = ops.zeros_acc(B, device=dev) if want_lv else None
        check(lib().swiftk_trigflow_loss(Fx.data_ptr(), vt.data_ptr(), None if lv is None else lv.contiguous().data_ptr(),
                                         wv.data_ptr(), wl.data_ptr(), loss.data_ptr(), dF.data_ptr(),
                                         None if dlv is None else dlv.data_ptr(), sd, B, C, H, W, 1.0, st), "swiftk_trigflow_loss")

        class Runner:
            value = loss.reshape(())

            @staticmethod
            def run_backward(g):
                dF.mul_(g)
                if dlv is not None:
                    dlv.mul_(g)
                eng.backward(ctx, dF, dlv, grads_final=getattr(net, "reduce_params", None))

        return _Deferred.apply(self._anchor(dev), Runner)


class SCMLoss(_LossBase):
    """Continuous-time consistency (sCM) loss, loss.py:163-260, with the tangent from ``jvp_engine.SwinJvpEngine``.

    Same constructor kwargs and call signature as the reference.  Per iteration: one tangent pass (no gradient; it
    carries primal and tangent rows through every GEMM, ~2 forward passes of work), one forward-with-activations pass
    and its backward.  ``jvp_dtype``: operand type of the tangent pass (bf16 = the trainer's autocast; fp32 for
    parity checks against the fp32 oracle).
    """

    def __init__(self, dataset, noise: dict, sigma_data: float, tangent_warmup_kimg: int = 0, distillation: bool = False,
                 jvp_dtype: str = "bf16"):
        super().__init__(dataset, sigma_data)
        self.cfg = dict(noise)
        self._sampling_fn = partial(NOISE_SAMPLING_METHODS[self.cfg.pop("dist")], **self.cfg)
        self.tangent_warmup_kimg = tangent_warmup_kimg
        self.distillation = distillation
        self.jvp_dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32}[jvp_dtype]

    def _jvp_engine(self, mod):
        from ..jvp_engine import SwinJvpEngine
        eng = getattr(mod.model, "_jvp_engine", None)
        if eng is None or eng.dt != self.jvp_dtype:
            eng = SwinJvpEngine(mod.model, self.jvp_dtype)
            mod.model._jvp_engine = eng
        return eng

    def forward(self, net, x, step, condition=None, auxiliary=None, net_pretrained=None, _tau=None, _z=None, **kwargs):
        mod = getattr(net, "module", net)
        eng = _engine(net)
        dev = x.device
        B, C, H, W = x.shape
        per = C * H * W
        sd = float(self.sigma_data)
        tau = self._sampling_fn(x) if _tau is None else _tau
        t = torch.atan(tau.reshape(B).float() / sd).contiguous()
        z = (torch.randn_like(x) if _z is None else _z).contiguous().float()
        x = x.contiguous().float()
        xt, dxt = torch.empty_like(x), torch.empty_like(x)  # x_t / sigma_d and dx_t/dt = cos t z sd - sin t x
        st = torch.cuda.current_stream().cuda_stream
        check(lib().swiftk_trigflow_prep(x.data_ptr(), z.data_ptr(), t.data_ptr(), xt.data_ptr(), dxt.data_ptr(), sd, B, per, st),
              "swiftk_trigflow_prep")
        aux = _process_auxiliary(auxiliary, mod.auxiliary_dim, B, dev)
        srcs = [xt]
        if condition is not None and mod.condition_channels > 0:
            srcs.append(condition.contiguous().float())
        if self.distillation and net_pretrained is not None:  # v-prediction teacher (loss.py:204-208)
            with torch.no_grad():
                dxt = (sd * net_pretrained(xt, t, condition, auxiliary)).float().contiguous()
        # tangent direction (loss.py:215-216): v_x = cos t sin t dxt / sd, v_t = cos t sin t
        cs = (torch.cos(t) * torch.sin(t)).contiguous()
        vx = torch.empty_like(x)
        check(lib().swiftk_axpby_per_sample(vx.data_ptr(), (cs / sd).contiguous().data_ptr(), dxt.data_ptr(), None, None, B, per,
                                            st), "swiftk_axpby_per_sample")
        want_lv = mod.model.logvar_embed is not None
        jeng = self._jvp_engine(mod)
        eng.refresh()
        jeng.refresh()
        # The reference runs the network twice here: torch.func.jvp for the tangent, then a grad-enabled forward
        # (loss.py:212-237).  The tangent pass's primal rows are that forward: with bf16 activations (the trainer's
        # autocast) they are kept per layer and handed to the backward pass -- one forward-equivalent of five saved.
        # (the kept buffers hold tangent rows too: 2 x the activations of a forward, 110 GiB peak at local batch 8 -- taken
        # when 4.4 activation sets fit into 75 % of the device, i.e. up to local batch 15 on 288 GB)
        env = os.environ.get("SWIFTK_SCM_ONE_PASS")
        fits = _plan_once(self._plans, ("scm_one_pass", B, env), lambda: int(env != "0") if env is not None else
                          int(4.4 * eng.activation_bytes(B) <= _memory_budget(dev)), "one-pass sCM", dev)
        one_pass = bool(jeng.dt == torch.bfloat16 and jeng.mlp_e == eng.mlp_e and fits)
        self.last_one_pass = one_pass
        with torch.no_grad():
            if one_pass:
                dF, Fx, lv, ctx = jeng.jvp(srcs, vx, t, cs, aux, save_ctx=True, want_logvar=want_lv)
            else:
                dF = jeng.jvp(srcs, vx, t, cs, aux)
        if not one_pass:
            res = eng.forward(srcs, [1.0] * len(srcs), t, aux, want_logvar=want_lv)
            Fx, lv, ctx = res if want_lv else (res[0], None, res[1])
        r = min(1.0, step / (self.tangent_warmup_kimg * 1000)) if self.tangent_warmup_kimg > 0 else 1.0
        target = torch.empty_like(Fx)
        ss = torch.empty(B, device=dev)
        check(lib().swiftk_scm_target(Fx.data_ptr(), dxt.data_ptr(), xt.data_ptr(), dF.contiguous().data_ptr(), t.data_ptr(),
                                      float(r), sd, target.data_ptr(), ss.data_ptr(), B, per, st), "swiftk_scm_target")
        # (F - F.detach() - g)^2 == (1 * F - target)^2 with target = F.detach() + g held constant
        wv, wl = self._w(dev)
        loss = ops.zeros_acc(1, device=dev)  # (atomically accumulated: cleared by the library's own kernel)
        dFx = torch.empty_like(Fx)
        dlv = ops.zeros_acc(B, device=dev) if want_lv else None
        check(lib().swiftk_trigflow_loss(Fx.data_ptr(), target.data_ptr(), None if lv is None else lv.contiguous().data_ptr(),
                                         wv.data_ptr(), wl.data_ptr(), loss.data_ptr(), dFx.data_ptr(),
                                         None if dlv is None else dlv.data_ptr(), 1.0, B, C, H, W, 1.0, st), "swiftk_trigflow_loss")

        class Runner:
            value = loss.reshape(())

            @staticmethod
            def run_backward(g):
                dFx.mul_(g)
                if dlv is not None:
                    dlv.mul_(g)
                eng.backward(ctx, dFx, dlv, grads_final=getattr(net, "reduce_params", None))

        self._last = dict(dF=dF, Fx=Fx)  # kept for tests / diagnostics
        return _Deferred.apply(self._anchor(dev), Runner)


class CRPSLoss(_LossBase):
    """Multistep almost-fair CRPS (loss.py:306-445): ensemble_size rollouts of `steps` network calls each."""

    def __init__(self, dataset, sigma_data: float, ensemble_size: int = 2, alpha: float = 1.0):
        super().__init__(dataset, sigma_data)
        self.ensemble_size, self.alpha = ensemble_size, alpha

    @staticmethod
    def _n_keep(eng, B, calls, dev) -> int:
        """How many trailing network calls keep their activations: SWIFTK_CRPS_KEEP if set.  ALL of them when the iteration
        then fits into 85 % of the device (no recompute slot is needed in that case; measured working set beside the kept
        sets: 0.75 activation sets, planned as 1.2 -- BASELINE configs[4] at local batch 8: 8 of 8, 218 GiB peak of 288).
        Otherwise what fits into 75 % beside 2.2 activation sets (rollout states, the recompute slot, one step's backward
        temporaries, gradients, optimiser state): 2 at batch 16, none at batch 32."""
        env = os.environ.get("SWIFTK_CRPS_KEEP")
        if env is not None:
            return max(0, min(calls, int(env)))
        act = max(1, eng.activation_bytes(B))
        if (calls + 1.2) * act <= _memory_budget(dev, 0.85):
            return calls
        return max(0, min(calls, int(_memory_budget(dev) / act - 2.2)))

    def _forcings_all(self, idx, aux_host, steps, dev):
        """Standardised forcings of every rollout step [steps, B, n_forc, H, W] on the device (loss.py:378-392 reads them per
        sample and step inside ``_one_step``).  ONE asynchronous upload out of a pinned staging buffer: a copy from pageable
        memory makes the host wait for everything queued before it -- the whole previous iteration -- and the GPU then idles
        while the host prepares this one (r04 trace: 107 ms of an 800 ms iteration)."""
        rows = [torch.stack([self.dataset.get_forcings(int(j) + int(i * float(dt) * 10 // 6)) for j, dt in zip(idx, aux_host)], 0)
                for i in range(steps)]
        host = self.dataset.standardize_x(torch.stack(rows, 0).flatten(0, 1)).float().reshape(steps, len(idx), *rows[0].shape[1:])
        if torch.device(dev).type != "cuda":
            return host.to(dev)
        ring = self.__dict__.setdefault("_forc_ring", {"slots": [], "n": 0})
        if len(ring["slots"]) < 2:
            ring["slots"].append([None, None])
        slot = ring["slots"][ring["n"] % 2]
        ring["n"] += 1
        if slot[0] is None or slot[0].shape != host.shape:
            slot[0], slot[1] = torch.empty(host.shape, dtype=torch.float32, pin_memory=True), None
        if slot[1] is not None:
            slot[1].synchronize()  # the upload issued from this buffer two iterations ago (long finished)
        slot[0].copy_(host)
        out = slot[0].to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return out

    def forward(self, net, target, condition, auxiliary, idx, steps: int = 1, chunk_size: int = 2, _latents=None, **kwargs):
        mod = getattr(net, "module", net)
        eng = _engine(net)
        dev = target.device
        B, C, H, W = target.shape
        sd = float(self.sigma_data)
        nv = len(self.dataset.variables)
        # per-sample lead times and indices as host numbers, read ONCE (the trainer hands them over on the CPU; a device
        # tensor costs one stall here instead of one per sample and step as in loss.py:378-392)
        aux_host = [float(v) for v in (auxiliary.tolist() if torch.is_tensor(auxiliary) else auxiliary)]
        idx = [int(j) for j in (idx.tolist() if torch.is_tensor(idx) else idx)]
        delta = int(aux_host[0] * 10)  # NOTE: assumes same delta within a batch (loss.py:378)
        mx, sx, stt = self.dataset.rollout_stats(delta, dev)
        coef = (stt / sx).contiguous()         # cond_std += pred * st/sx  ==  standardize(unstd(cond) + unstd_t(pred))
        aux = _process_auxiliary(auxiliary, mod.auxiliary_dim, B, dev)
        t = torch.full((B,), math.pi / 2, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        hw = H * W
        forc = self._forcings_all(idx, aux_host, steps, dev)
        target = target.contiguous().float()
        E = self.ensemble_size
        lat = [[(torch.randn_like(target) if _latents is None else _latents[e][i].to(dev)).contiguous() for i in range(steps)]
               for e in range(E)]
        conds = [[None] * steps for _ in range(E)]
        preds = torch.empty(E, B, C, H, W, device=dev)
        # ---- pass 1: the rollouts on the inference engine (bf16 operands), saving each step's input state.  The reference
        # recomputes every step in the backward pass (checkpoint_sequential, loss.py:395-437) because activations of 2 x steps
        # network calls do not fit its GPUs; with 288 GB of HBM the LAST `n_keep` calls (in rollout order: the ones pass 2
        # differentiates first) run on the training engine and keep their activations, one buffer set ("slot") each
        n_keep = self.last_n_keep = _plan_once(self._plans, ("crps_keep", B, E * steps, os.environ.get("SWIFTK_CRPS_KEEP")),
                                               lambda: self._n_keep(eng, B, E * steps, dev), "resident rollout steps", dev)
        kept = {}
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for e in range(E):
                cond = condition[:, :nv].contiguous().float()
                for i in range(steps):
                    conds[e][i] = cond
                    c = e * steps + i
                    if c >= E * steps - n_keep:
                        kept[(e, i)] = eng.forward([lat[e][i], cond, forc[i]], [1.0, 1.0, 1.0], t, aux, slot=1 + E * steps - 1 - c)
                        out = kept[(e, i)][0]
                    else:
                        out = mod(lat[e][i], t, (cond, forc[i]), aux, x_scale=1.0)  # x_t / sigma_d with x_t = z * sigma_d
                    if i < steps - 1:
                        nxt = torch.empty_like(cond)
                        # pred = -sigma_d * out ; cond' = cond + pred * st/sx
                        check(lib().swiftk_channel_axpy(nxt.data_ptr(), cond.data_ptr(), out.data_ptr(), (-sd * coef).data_ptr(),
                                                        B, C, hw, st), "swiftk_channel_axpy")
                        cond = nxt
                    else:
                        ops.axpby(-sd, out, 0.0, out, out=preds[e])
        wv, wl = self._w(dev)
        loss = ops.zeros_acc(1, device=dev)  # (atomically accumulated: cleared by the library's own kernel)
        dpreds = torch.empty_like(preds)
        check(lib().swiftk_crps_loss(preds.data_ptr(), target.data_ptr(), wv.data_ptr(), wl.data_ptr(), loss.data_ptr(),
                                     dpreds.data_ptr(), E, B, C, H, W, float(self.alpha), 1.0, st), "swiftk_crps_loss")

        def run_backward(g):
            # ---- pass 2: per member, walk the steps backwards; recompute one step with activations, then backprop it
            for e in reversed(range(E)):
                gcond = None  # dL/d cond_{i+1}
                for i in reversed(range(steps)):
                    if (e, i) in kept:
                        out, ctx = kept.pop((e, i))  # (a slot's buffers stay valid until its next replay: the next iteration)
                    else:
                        out, ctx = eng.forward([lat[e][i], conds[e][i], forc[i]], [1.0, 1.0, 1.0], t, aux)
                    dout = torch.empty_like(out)
                    if i == steps - 1:
                        ops.axpby(-sd, dpreds[e], 0.0, dpreds[e], out=dout)
                        dout.mul_(g)
                    else:
                        check(lib().swiftk_channel_axpy(dout.data_ptr(), None, gcond.data_ptr(), (-sd * coef).data_ptr(), B, C, hw,
                                                        torch.cuda.current_stream().cuda_stream), "swiftk_channel_axpy")
                    need = i > 0
                    last = (e == 0 and i == 0)  # the final backward pass of the iteration: gradients complete layer by layer
                    dins = eng.backward(ctx, dout, None, need_input_grad=(False, need, False),
                                        grads_final=getattr(net, "reduce_params", None) if last else None)
                    if need:  # (a replayed backward returns ITS OWN output tensor, overwritten by the next replay: copy it)
                        gcond = dins[1].clone() if gcond is None else ops.axpby(1.0, gcond, 1.0, dins[1])

        class Runner:
            value = loss.reshape(())

        Runner.run_backward = staticmethod(run_backward)
        return _Deferred.apply(self._anchor(dev), Runner)
