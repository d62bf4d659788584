"""Forward-mode tangent of the SwinV2 denoiser on the gfx950 kernels: (x, t; dx, dt) -> dF.

What the reference obtains with ``torch.func.jvp(lambda x, t: net.module(x, t, cond, aux, jvp=True), ...)``
(training/loss.py:212-220; ``jvp=True`` selects the explicit ``softmax(q k^T) v`` attention, swinv2.py:129-133).
Here the tangent is propagated explicitly, one kernel per step:

  * linear maps (patch embedding, to_qkv, wo, w1, w2, head, the small latent/modulation linears): the tangent rides
    as extra rows of the same GEMM -- primal rows 0..M-1, tangent rows M..2M-1 of one ``swiftk_gemm`` launch, so the
    weight panels are streamed once for both;
  * non-linear steps: ``swiftk_timestep_embed_jvp``, ``swiftk_silu_jvp``, ``swiftk_qknorm_jvp``,
    ``swiftk_window_attention_jvp``, ``swiftk_modnorm_jvp``, ``swiftk_swiglu_jvp`` (csrc/jvp_kernels.hip).

The condition and auxiliary inputs carry no tangent (loss.py:212-213 closes over them).  ``dtype`` is the storage /
GEMM-operand type of the activations (bf16 under the trainer's autocast, fp32 for parity checks); residual stream,
normalisation statistics and the attention products are fp32 either way.  No parameter gradients: the sCM loss detaches
the tangent (loss.py:238-240).
"""
from __future__ import annotations

import math
import os
from typing import Optional, Sequence

import torch

from . import ops
from .graphs import GraphCache
from ._lib import EPI_BIAS_POS, EPI_NONE, EPI_QKNORM_JVP, EPI_SWIGLU_JVP, SwiftkError, check, lib


def _s():
    return torch.cuda.current_stream().cuda_stream


def _gemm(a, w, out, epi=EPI_NONE, ep0=None, ep1=None, pos_rows=0, k=None):
    M, K = a.shape
    # `k`: the operand's valid width when it ends half-way into the last 64-deep k-tile (d = 1056 = 16.5 tiles): the kernel then
    # skips the zero half (both operands' rows extend to the end of the tile)
    if k is not None and k % 64 == 32 and K >= k + 32 and w.stride(0) >= k + 32:
        K = k
    check(lib().swiftk_gemm(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, w.shape[0],
                            K, ops.dtype_code(a.dtype), ops.dtype_code(out.dtype), epi, None if ep0 is None else ep0.data_ptr(),
                            None if ep1 is None else ep1.data_ptr(), pos_rows, _s()), "swiftk_gemm")
    return out


class SwinJvpEngine:
    def __init__(self, module, dtype: torch.dtype = torch.bfloat16):
        self.m = module
        self.hd = module.dim // module.heads
        if self.hd not in (80, 88, 96):
            raise SwiftkError("the tangent kernels are built for head_dim 80 / 88 / 96")
        self.dt = dtype
        self._stamp = None
        self._buf = {}            # persistent operand copies (fixed addresses for the captured launch sequence)
        self.graphs = GraphCache()

    def refresh(self):
        m, dt = self.m, self.dt
        stamp = tuple((p.data_ptr(), p._version) for p in m.parameters())
        # the GEMM operands may be the TRAINING engine's (below): their addresses are part of what a captured sequence baked in,
        # so the engine's identity and its operand addresses belong to the stamp -- a recreated training engine, a
        # reallocated operand buffer or ``net.to(...)`` would otherwise leave the replays reading freed memory
        te = getattr(m, "_train_engine", None)
        te_key = None
        if te is not None and dt == torch.bfloat16:
            te.refresh()
            te_key = (id(te),) + tuple(l[k].data_ptr() for l in te.L for k in ("qkv", "wo", "w1", "w2"))
        if stamp == self._stamp and te_key == getattr(self, "_te_key", None):
            return
        if te_key != getattr(self, "_te_key", None) and getattr(self, "_share", False):
            self.graphs.invalidate()
        self._te_key = te_key
        ptrs = tuple(p.data_ptr() for p in m.parameters())  # captured sequences read some parameters in place
        if ptrs != getattr(self, "_ptrs", None):
            if getattr(self, "_ptrs", None) is not None:
                self.graphs.invalidate()
            self._ptrs = ptrs
        d, mlp = m.dim, m.mlp_dim
        mlp_e = self.mlp_e = (mlp + 7) // 8 * 8  # odd MLP widths (dim 1280 -> 3413): zero (gate, up) row pairs, as in SwinEngine
        self.kd, self.kmlp = ops.k_pad(dt, d), ops.k_pad(dt, mlp_e)
        self.kpe = ops.k_pad(dt, m.in_channels * m.patch_size[0] * m.patch_size[1])
        dev0 = m.pos_embed.device
        ctr = [0]

        def keep(name, make, fill):
            if name not in self._buf:
                self._buf[name] = make()
            fill(self._buf[name])
            return self._buf[name]

        def cast(w, k):
            ctr[0] += 1
            w = w.detach()
            return keep(f"c{ctr[0]}", lambda: torch.empty(w.shape[0], k, dtype=dt, device=dev0),
                        lambda b: ops.pad_cols(w, k, dt, out=b))

        def keep_f32(t):
            ctr[0] += 1
            t = t.detach().float()
            return keep(f"f{ctr[0]}", lambda: torch.empty_like(t, memory_format=torch.contiguous_format), lambda b: b.copy_(t))

        # the sCM loss runs this engine's pass and the training engine's backward on the same weights (loss.py:212-237): with bf16
        # operands and an MLP width both engines pad alike, the four GEMM operands per layer are the training engine's (one cast
        # per optimizer step instead of two, and no second (gate, up) interleave of w1)
        share = te is not None and dt == torch.bfloat16 and getattr(te, "mlp_e", None) == mlp_e
        if share:
            share = te.kd == self.kd and te.kmlp == self.kmlp and len(te.L) == len(m.transformer.layers)
        if getattr(self, "_share", share) != share:
            self.graphs.invalidate()  # (captured sequences hold the other set of operand addresses)
        self._share = share
        def keep_cat(name, ts):  # fp32 concatenation written straight into its persistent buffer
            ts = [t.float() for t in ts]
            if name not in self._buf:
                self._buf[name] = torch.cat(ts, 0)
            else:
                torch.cat(ts, 0, out=self._buf[name])
            return self._buf[name]

        self.L = []
        mods_w, mods_b = [], []
        for li, (att, ff) in enumerate(m.transformer.layers):
            if share:
                ops4 = {k: te.L[li][k] for k in ("qkv", "wo", "w1", "w2")}
            else:
                w1i = ff.w1.weight.detach().view(2, mlp, d).permute(1, 0, 2).reshape(2 * mlp, d)  # (gate_j, up_j) interleaved
                if mlp_e != mlp:
                    w1i = torch.cat([w1i, w1i.new_zeros(2 * (mlp_e - mlp), d)], 0)
                ops4 = dict(qkv=cast(att.to_qkv.weight, self.kd), wo=cast(att.wo.weight, self.kd), w1=cast(w1i, self.kd),
                            w2=cast(ff.w2.weight, self.kmlp))
            self.L.append(dict(**ops4, scale=att.scale.detach().reshape(-1).float().contiguous(),
                               g1=att.norm.norm.weight.detach().float().contiguous(),
                               b1=att.norm.norm.bias.detach().float().contiguous(),
                               g2=ff.norm.norm.weight.detach().float().contiguous(),
                               b2=ff.norm.norm.bias.detach().float().contiguous()))
            mods_w += [att.norm.modulation.weight.detach(), ff.norm.modulation.weight.detach()]
            mods_b += [att.norm.modulation.bias.detach(), ff.norm.modulation.bias.detach()]
        self.pe = cast(m.patch_embed.emb.weight, self.kpe)
        hw = m.head.head[0].weight.detach()
        if hw.shape[0] % 4:  # GEMM N granularity (1x1 patches: 69 -> 72 output columns, the zero ones unused)
            hw = torch.cat([hw, hw.new_zeros(4 - hw.shape[0] % 4, hw.shape[1])], 0)
        self.head = cast(hw, self.kd)
        self.mod_w, self.mod_b = keep_cat("mod_w", mods_w), keep_cat("mod_b", mods_b)
        half = d // 2
        if "freqs" not in self._buf:
            self._buf["freqs"] = torch.exp(-math.log(10_000) * torch.arange(half, dtype=torch.float32) / half).to(dev0)
        self.freqs = self._buf["freqs"]
        self._stamp = stamp

    @staticmethod
    def _f(p):
        return p.detach().float().contiguous()

    def jvp(self, srcs: Sequence[torch.Tensor], dsrc0: torch.Tensor, t: torch.Tensor, dt_: torch.Tensor,
            aux: Optional[torch.Tensor], save_ctx: bool = False, want_logvar: bool = False):
        """srcs: channel-concatenated network inputs (srcs[0] carries the tangent ``dsrc0``); t, dt_: [B].  Returns dF
        (after the first call of a signature: the captured sequence's own tensor, overwritten by the next call).

        ``save_ctx``: the primal rows of the pass ARE a forward pass; keep them (per-layer buffers instead of one reused set)
        and return ``(dF, F, logvar, ctx)`` with ``ctx`` in the form ``SwinTrainEngine.backward`` takes -- the sCM loss
        then needs no second forward pass through the network (loss.py:212-237 runs ``jvp`` and a grad-enabled forward)."""
        self.refresh()
        srcs = [s.contiguous().float() for s in srcs]
        ins = list(srcs) + [dsrc0.contiguous().float(), t.contiguous().float(), dt_.contiguous().float()] + \
            ([aux.contiguous().float()] if aux is not None else [])
        n = len(srcs)
        key = ("jvp", tuple(tuple(s.shape) for s in srcs), aux is not None, bool(save_ctx), bool(want_logvar))
        fn = lambda *a: self._jvp(list(a[:n]), a[n], a[n + 1], a[n + 2], a[n + 3] if aux is not None else None, save_ctx, want_logvar)
        res = self.graphs.call(key, fn, ins)
        if save_ctx:
            res[3]["graph_key"] = (key, max(1, self.graphs.generation.get(key, 0)))  # eager runs before the first capture count as generation 1
        return res

    def _jvp(self, srcs: Sequence[torch.Tensor], dsrc0: torch.Tensor, t: torch.Tensor, dt_: torch.Tensor,
             aux: Optional[torch.Tensor], save: bool = False, want_logvar: bool = False):
        m, T = self.m, self.dt
        tc = ops.dtype_code(T)
        dev = srcs[0].device
        B = srcs[0].shape[0]
        d, heads, mlp = m.dim, m.heads, self.mlp_e
        gh, gw = m.grid_size
        ntok = gh * gw
        M = B * ntok
        L = lib()
        srcs = [s.contiguous().float() for s in srcs]
        dsrc0 = dsrc0.contiguous().float()
        t, dt_ = t.contiguous().float(), dt_.contiguous().float()
        # ---- time embedding -> latent -> modulation, and their tangents
        aux_s = aux.contiguous().float() if (m.auxiliary_embed is not None and aux is not None) else None
        emb = ops.timestep_embed(t, aux_s, self.freqs, None if aux_s is None else self._f(m.auxiliary_embed.weight),
                                 None if aux_s is None else self._f(m.auxiliary_embed.bias), d, float(m.timestep_weight))
        demb = torch.empty_like(emb)
        check(L.swiftk_timestep_embed_jvp(t.data_ptr(), dt_.data_ptr(), self.freqs.data_ptr(), demb.data_ptr(), B, d,
                                          float(m.timestep_weight), _s()), "swiftk_timestep_embed_jvp")

        def silu_pair(z, dz):
            y, dy = torch.empty_like(z), torch.empty_like(z)
            check(L.swiftk_silu_jvp(z.data_ptr(), dz.data_ptr(), y.data_ptr(), dy.data_ptr(), z.numel(), _s()), "swiftk_silu_jvp")
            return y, dy

        l1w, l1b = self._f(m.latent_embed.l1.weight), self._f(m.latent_embed.l1.bias)
        l2w, l2b = self._f(m.latent_embed.l2.weight), self._f(m.latent_embed.l2.bias)
        z1 = ops.linear_small(emb, l1w, l1b, 0)
        h1, dh1 = silu_pair(z1, ops.linear_small(demb, l1w, None, 0))
        z2 = ops.linear_small(h1, l2w, l2b, 0)
        lat, dlat = silu_pair(z2, ops.linear_small(dh1, l2w, None, 0))
        mod = ops.linear_small(lat, self.mod_w, self.mod_b, 0)      # [B, depth * 2 * 2d]
        dmod = ops.linear_small(dlat, self.mod_w, None, 0)
        ldmod = mod.stride(0)
        # ---- patch embedding: primal with bias + pos_embed, tangent without
        rest = sum(s.shape[1] for s in srcs[1:])
        ape = ops.patchify(srcs, [1.0] * len(srcs), m.patch_size, self.kpe, T)
        dsr = [dsrc0] + ([torch.zeros(B, rest, *dsrc0.shape[2:], device=dev)] if rest else [])
        dape = ops.patchify(dsr, [1.0] * len(dsr), m.patch_size, self.kpe, T)
        X = torch.empty(2 * M, d, dtype=torch.float32, device=dev)   # residual stream: primal rows, then tangent rows
        x, dx = X[:M], X[M:]
        _gemm(ape, self.pe, x, EPI_BIAS_POS, self._f(m.patch_embed.emb.bias), self._f(m.pos_embed).reshape(ntok, d), ntok)
        _gemm(dape, self.pe, dx)
        es = torch.empty(0, dtype=T).element_size()

        def operand(width, valid):  # [2M, width] GEMM operand, k-padding columns zero
            b = torch.empty(2 * M, width, dtype=T, device=dev)
            if width > valid:
                b[:, valid:].zero_()
            return b

        XT_in = operand(self.kd, d)
        # bf16 operands: the stream and its tangent as (bf16 hi, 8-bit lo) pairs, hi = the operand rows of XT (16 instead of 24
        # bytes per element in the norm kernel); fp32 operands keep the fp32 stream
        pair = T == torch.bfloat16 and d % 8 == 0 and d <= 1536 and not os.environ.get("SWIFTK_TRAIN_FP32_STREAM")
        if pair:
            XLO = torch.empty(2 * M, d, dtype=torch.uint8, device=dev)
            check(L.swiftk_split_pair(X.data_ptr(), d, XT_in.data_ptr(), self.kd, XLO.data_ptr(), d, 8, 2 * M, d, _s()), "swiftk_split_pair")
        else:
            check(L.swiftk_cast_pad(X.data_ptr(), d, XT_in.data_ptr(), self.kd, 2 * M, d, tc, _s()), "swiftk_cast_pad")
        # to_qkv and w1 with the non-linear step behind them (QK normalisation, SwiGLU) and its tangent rule in the GEMM's epilogue:
        # the persistent kernel pairs every primal row with its tangent row inside one lane (swiftk_gemm_jvp); the separate
        # swiftk_qknorm_jvp / swiftk_swiglu_jvp passes (and the raw products they re-read) remain for fp32 operands
        fused = (T == torch.bfloat16 and M % 128 == 0 and d % 64 in (0, 32) and self.hd in (80, 88, 96) and heads % 2 == 0
                 and mlp % 8 == 0 and not os.environ.get("SWIFTK_JVP_UNFUSED"))
        kk = d  # (d = 16.5 k-tiles: the kernel skips the zero half of the last one)
        # one buffer set reused by every layer -- or, when the primal rows are kept for a backward pass, one set per layer
        shared = None if save else dict(QKV=torch.empty(2 * M, 3 * d, dtype=T, device=dev), ATT=operand(self.kd, d),
                                        Y=torch.empty(2 * M, d, dtype=T, device=dev),
                                        H=None if fused else torch.empty(2 * M, 2 * mlp, dtype=T, device=dev), HM=operand(self.kmlp, mlp))

        def modnorm(i2, gamma, beta, Y, XT, XT_prev):
            off = i2 * 2 * d * 4
            if pair:
                check(L.swiftk_modnorm_jvp_pair(Y.data_ptr(), Y.data_ptr() + M * d * es, d, XT_prev.data_ptr(),
                                                XT_prev.data_ptr() + M * self.kd * es, XT.data_ptr(), XT.data_ptr() + M * self.kd * es,
                                                self.kd, XLO.data_ptr(), XLO.data_ptr() + M * d, gamma.data_ptr(), beta.data_ptr(),
                                                mod.data_ptr() + off, dmod.data_ptr() + off, ldmod, M, d, ntok, 1e-6, _s()),
                      "swiftk_modnorm_jvp_pair")
                return
            check(L.swiftk_modnorm_jvp(Y.data_ptr(), Y.data_ptr() + M * d * es, d, x.data_ptr(), dx.data_ptr(), XT.data_ptr(),
                                       XT.data_ptr() + M * self.kd * es, self.kd, gamma.data_ptr(), beta.data_ptr(),
                                       mod.data_ptr() + off, dmod.data_ptr() + off, ldmod, M, d, ntok, 1e-6, tc, _s()),
                  "swiftk_modnorm_jvp")

        layers = []
        do_shift = any(m.shift_size)
        for i in range(len(self.L)):
            W = self.L[i]
            sh = tuple(m.shift_size) if (do_shift and i % 2) else (0, 0)
            QKV = shared["QKV"] if shared else torch.empty(2 * M, 3 * d, dtype=T, device=dev)
            rn = torch.empty(M, 3 * heads, dtype=torch.float32, device=dev) if save else None
            if fused:  # to_qkv with the cosine-attention prologue and its tangent in the GEMM's epilogue (paired rows)
                check(L.swiftk_gemm_jvp(XT_in.data_ptr(), XT_in.stride(0), W["qkv"].data_ptr(), W["qkv"].stride(0), QKV.data_ptr(), 3 * d,
                                        M, 3 * d, kk, EPI_QKNORM_JVP, W["scale"].data_ptr(),
                                        None if rn is None else rn.data_ptr(), self.hd, None, 0, _s()), "swiftk_gemm_jvp")
            else:
                _gemm(XT_in, W["qkv"], QKV, k=d)
                check(L.swiftk_qknorm_jvp(QKV.data_ptr(), QKV.data_ptr() + M * 3 * d * es, 3 * d, W["scale"].data_ptr(),
                                          None if rn is None else rn.data_ptr(), M, heads, self.hd, tc, _s()), "swiftk_qknorm_jvp")
            ATT = shared["ATT"] if shared else operand(self.kd, d)
            check(L.swiftk_window_attention_jvp(QKV.data_ptr(), QKV.data_ptr() + M * 3 * d * es, 3 * d, ATT.data_ptr(),
                                                ATT.data_ptr() + M * self.kd * es, self.kd, B, gh, gw, heads, self.hd, sh[0], sh[1],
                                                tc, _s()), "swiftk_window_attention_jvp")
            Y1 = shared["Y"] if shared else torch.empty(2 * M, d, dtype=T, device=dev)
            _gemm(ATT, W["wo"], Y1, k=d)
            XT_mid = XT_in if shared else operand(self.kd, d)
            modnorm(2 * i, W["g1"], W["b1"], Y1, XT_mid, XT_in)
            HM = shared["HM"] if shared else operand(self.kmlp, mlp)
            if fused:  # w1 with the gate and its tangent in the epilogue; the primal pre-activations are kept only for a backward pass
                H = torch.empty(M, 2 * mlp, dtype=T, device=dev) if save else None
                check(L.swiftk_gemm_jvp(XT_mid.data_ptr(), XT_mid.stride(0), W["w1"].data_ptr(), W["w1"].stride(0),
                                        None if H is None else H.data_ptr(), 2 * mlp, M, 2 * mlp, kk,
                                        EPI_SWIGLU_JVP, None, None, 0, HM.data_ptr(), self.kmlp, _s()), "swiftk_gemm_jvp")
            else:
                H = shared["H"] if shared else torch.empty(2 * M, 2 * mlp, dtype=T, device=dev)
                _gemm(XT_mid, W["w1"], H, k=d)
                check(L.swiftk_swiglu_jvp(H.data_ptr(), H.data_ptr() + M * 2 * mlp * es, 2 * mlp, HM.data_ptr(),
                                          HM.data_ptr() + M * self.kmlp * es, self.kmlp, M, mlp, tc, _s()), "swiftk_swiglu_jvp")
            Y2 = shared["Y"] if shared else torch.empty(2 * M, d, dtype=T, device=dev)
            _gemm(HM, W["w2"], Y2)
            XT_out = XT_mid if shared else operand(self.kd, d)
            modnorm(2 * i + 1, W["g2"], W["b2"], Y2, XT_out, XT_mid)
            if save:
                layers.append(dict(xT_in=XT_in[:M], qkvh=QKV[:M], rn=rn, att=ATT[:M], y1=Y1[:M], xT_mid=XT_mid[:M], h=H[:M],
                                   hmid=HM[:M], y2=Y2[:M], shift=sh))
            XT_in = XT_out
        po4 = self.head.shape[0]
        shape = (B, m.out_channels, *m.image_size)
        if not save:
            tok = torch.empty(M, po4, dtype=torch.float32, device=dev)
            _gemm(XT_in[M:], self.head, tok, k=d)
            return ops.unpatchify_affine(tok.view(B, ntok, po4), shape, m.patch_size)
        tok = torch.empty(2 * M, po4, dtype=torch.float32, device=dev)
        _gemm(XT_in, self.head, tok, k=d)
        F = ops.unpatchify_affine(tok[:M].view(B, ntok, po4), shape, m.patch_size)
        dF = ops.unpatchify_affine(tok[M:].view(B, ntok, po4), shape, m.patch_size)
        logvar = None
        if want_logvar:
            logvar = ops.linear_small(lat, self._f(m.logvar_embed.weight), self._f(m.logvar_embed.bias), 0).reshape(B)
        ctx = dict(B=B, M=M, srcs_ch=[s_.shape[1] for s_ in srcs], scales=[1.0] * len(srcs), layers=layers, aux=aux_s, emb=emb,
                   z1=z1, h1=h1, z2=z2, lat=lat, mod=mod, ape=ape, xT_final=XT_in[:M])
        return dF, F, logvar, ctx
