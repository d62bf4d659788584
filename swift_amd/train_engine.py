"""Forward-with-activations and backward of the SwinV2 denoiser on the gfx950 kernels (training step).

The reference trains under ``torch.autocast(bfloat16)`` (training/trainer.py:189-197) and lets autograd differentiate
the ATen graph.  Here the backward pass is explicit: every product is a hand-written kernel called through the C ABI --
``swiftk_gemm`` for the data gradients (A = dY, operand = W^T copy), ``swiftk_gemm_splitk`` + ``swiftk_reduce_slabs``
for the weight gradients (A = dY^T, operand = X^T, contraction over all tokens split across workgroups),
``swiftk_window_attention_bwd`` / ``swiftk_qknorm_bwd`` / ``swiftk_modnorm_bwd`` / ``swiftk_swiglu_bwd`` for the rest.
Activations are saved per layer (3.4 GB per sample at Swift-B; the multistep loss recomputes one rollout step at a time,
as the reference's ``checkpoint_sequential`` does).  Parameter gradients accumulate into ``param.grad`` (fp32).

bf16 GEMM operands, fp32 accumulation, fp32 residual stream / normalisation statistics / gradients of the residual
stream -- the same precision split as the inference engine.  head_dim 80 / 88 / 96 (the 468 M, Swift-B and 664 M
variants of experiment/era5-swinv2-1.4-scm.yaml:21-36).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

import torch

from . import ops
from ._lib import (ATTN_PRENORM, BF16, EPI_ACCUM, EPI_BIAS_POS, EPI_NONE, EPI_QKNORM, EPI_SWIGLU_BOTH, EPI_SWIGLU_BWD, F32, SwiftkError, check, lib)
from .graphs import GraphCache

_BF = torch.bfloat16


def _s():
    return torch.cuda.current_stream().cuda_stream


def _gemm(a, w, out, epi=EPI_NONE, ep0=None, ep1=None, pos_rows=0, k=None):
    """out[M,N] = a[M,K] @ w[N,K]^T (row-major 2-D tensors, explicit strides)."""
    M, K = a.shape
    # `k`: the operand's valid width when it ends half-way into the last 64-deep k-tile (d = 1056 = 16.5 tiles): the kernel then
    # skips the zero half (both operands' rows extend to the end of the tile)
    if k is not None and k % 64 == 32 and K >= k + 32 and w.stride(0) >= k + 32:
        K = k
    check(lib().swiftk_gemm(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, w.shape[0],
                            K, ops.dtype_code(a.dtype), ops.dtype_code(out.dtype), epi, None if ep0 is None else ep0.data_ptr(),
                            None if ep1 is None else ep1.data_ptr(), pos_rows, _s()), "swiftk_gemm")
    return out


def _transpose(src, rows, cols, ldd=None):
    """[rows, >=cols] -> [cols, ldd] (ldd >= rows, zero padded)."""
    ldd = ldd or rows
    dst = torch.empty(cols, ldd, dtype=src.dtype, device=src.device)
    check(lib().swiftk_transpose(src.data_ptr(), src.stride(0), dst.data_ptr(), ldd, rows, cols, ops.dtype_code(src.dtype),
                                 _s()), "swiftk_transpose")
    return dst


def _zeros(*shape, device):
    """fp32 zeros that a kernel is about to ACCUMULATE into (atomics / read-modify-write), cleared by the library's own fill kernel
    on the current stream (``ops.zeros_acc``: one helper for every such site)."""
    return ops.zeros_acc(*shape, device=device)


def _padded(rows, width, valid):
    """[rows, width] bf16 GEMM operand whose columns [valid, width) are zero (the k-padding a GEMM may read); the kernel
    that fills it writes columns [0, valid) only -- zeroing 32 pad columns instead of the whole buffer."""
    t = torch.empty(rows, width, dtype=_BF, device=torch.device("cuda", torch.cuda.current_device()))
    if width > valid:
        t[:, valid:].zero_()
    return t


class SwinTrainEngine:
    def __init__(self, module):
        self.m = module
        self.hd = module.dim // module.heads
        if self.hd not in (80, 88, 96) or module.heads % 2:
            raise SwiftkError("the training kernels are built for head_dim 80 / 88 / 96 and an even head count")
        self._stamp = None
        self._slabs = None
        self._buf = {}            # persistent operand copies (fixed addresses: captured launch sequences read them)
        self.graphs = GraphCache()

    # ------------------------------------------------------------------ operands
    def _keep(self, name, make, fill):
        """Persistent buffer ``name``: allocated by ``make()`` once, refilled in place by ``fill(buf)`` on every refresh."""
        if name not in self._buf:
            self._buf[name] = make()
        fill(self._buf[name])
        return self._buf[name]

    def refresh(self):
        m = self.m
        stamp = tuple((p.data_ptr(), p._version) for p in m.parameters())
        if stamp == self._stamp:
            return
        self._check_addresses()
        d, mlp = m.dim, m.mlp_dim
        # int(8/3 dim) need not be a multiple of 4 (dim 1280 -> 3413): zero (gate, up) rows of w1 / zero columns of w2 bring
        # the MLP width to the GEMMs' granularity (N % 4, split-K rows % 8); silu(0) * 0 = 0 feeds w2's zero columns
        mlp_e = self.mlp_e = (mlp + 7) // 8 * 8
        self.kd, self.kmlp = ops.k_pad(_BF, d), ops.k_pad(_BF, mlp_e)
        self.kpe = ops.k_pad(_BF, m.in_channels * m.patch_size[0] * m.patch_size[1])
        self.kqkv = ops.k_pad(_BF, 3 * d)
        self.kh = ops.k_pad(_BF, 2 * mlp_e)
        po = m.out_channels * m.patch_size[0] * m.patch_size[1]
        self.kpo = ops.k_pad(_BF, po)
        dev0 = m.pos_embed.device
        ctr = [0]

        def cast(w, k):
            ctr[0] += 1
            w = w.detach()
            return self._keep(f"c{ctr[0]}", lambda: torch.empty(w.shape[0], k, dtype=_BF, device=dev0),
                              lambda b: ops.pad_cols(w, k, _BF, out=b))

        def tr(w, k):  # W^T operand for the data gradient
            return cast(w.detach().t().contiguous(), k)

        def both(w, k, kt, inter=0):
            """(forward operand [rows, k], data-gradient operand W^T [cols, kt]) of an fp32 weight in one pass over it;
            ``inter`` = mlp: rows leave (gate, up)-interleaved (``swiftk_cast_pad_t``)."""
            ctr[0] += 2
            w = w.detach()
            if w.dtype != torch.float32 or w.stride(1) != 1:
                w = w.float().contiguous()
            a = self._keep(f"c{ctr[0] - 1}", lambda: torch.empty(w.shape[0], k, dtype=_BF, device=dev0), lambda b: None)
            b_ = self._keep(f"c{ctr[0]}", lambda: torch.empty(w.shape[1], kt, dtype=_BF, device=dev0), lambda b: None)
            ops.cast_pad_t(w, a, b_, inter)
            return a, b_

        def keep_f32(t):
            ctr[0] += 1
            t = t.detach().float()
            return self._keep(f"f{ctr[0]}", lambda: torch.empty_like(t, memory_format=torch.contiguous_format),
                              lambda b: b.copy_(t))

        self.L = []
        for att, ff in m.transformer.layers:
            qkv, qkv_t = both(att.to_qkv.weight, self.kd, self.kqkv)
            wo, wo_t = both(att.wo.weight, self.kd, self.kd)
            if mlp_e != mlp:  # (the padded MLP width of dim 1280: interleave and zero-extend on the host side, then one pass each)
                w1i = ff.w1.weight.detach().view(2, mlp, d).permute(1, 0, 2).reshape(2 * mlp, d)
                w1i = torch.cat([w1i, w1i.new_zeros(2 * (mlp_e - mlp), d)], 0)
                w2p = torch.cat([ff.w2.weight.detach(), ff.w2.weight.new_zeros(d, mlp_e - mlp)], 1)
                w1, w1_t = both(w1i, self.kd, self.kh)
                w2, w2_t = both(w2p, self.kmlp, self.kd)
            else:
                w1, w1_t = both(ff.w1.weight, self.kd, self.kh, inter=mlp)
                w2, w2_t = both(ff.w2.weight, self.kmlp, self.kd)
            self.L.append(dict(qkv=qkv, qkv_t=qkv_t, wo=wo, wo_t=wo_t, w1=w1, w1_t=w1_t, w2=w2, w2_t=w2_t,
                               scale=att.scale.detach().reshape(-1).float().contiguous()))
        self.pe = cast(m.patch_embed.emb.weight, self.kpe)
        self.pe_t = tr(m.patch_embed.emb.weight, self.kd)
        hw = m.head.head[0].weight.detach()
        if hw.shape[0] % 4:  # GEMM N granularity (1x1 patches: 69 -> 72 output columns, the zero ones unused)
            hw = torch.cat([hw, hw.new_zeros(4 - hw.shape[0] % 4, hw.shape[1])], 0)
        self.head = cast(hw, self.kd)
        self.head_t = tr(m.head.head[0].weight, self.kpo)
        mods_w, mods_b = [], []
        for att, ff in m.transformer.layers:
            mods_w += [att.norm.modulation.weight.detach(), ff.norm.modulation.weight.detach()]
            mods_b += [att.norm.modulation.bias.detach(), ff.norm.modulation.bias.detach()]
        def keep_cat(name, ts):  # fp32 concatenation written straight into its persistent buffer
            ts = [t.float() for t in ts]
            if name not in self._buf:
                self._buf[name] = torch.cat(ts, 0)
            else:
                torch.cat(ts, 0, out=self._buf[name])
            return self._buf[name]

        self.mod_w, self.mod_b = keep_cat("mod_w", mods_w), keep_cat("mod_b", mods_b)
        half = d // 2
        if "freqs" not in self._buf:
            self._buf["freqs"] = torch.exp(-math.log(10_000) * torch.arange(half, dtype=torch.float32) / half).to(dev0)
        self.freqs = self._buf["freqs"]
        self._stamp = stamp

    # ------------------------------------------------------------------ helpers
    def _small(self, x, w, b, act=0):
        return ops.linear_small(x, w.detach().float().contiguous(), None if b is None else b.detach().float().contiguous(), act)

    def _wgrad(self, dy, x, rows, cols, out_grad, accumulate=True):
        """out_grad[rows, cols] (+)= dy[M, :rows]^T @ x[M, :cols] via split-K fp32 slabs; both operands token-major as the
        passes leave them (``swiftk_gemm_tn_splitk``); anything that kernel rejects (rows too short to read whole
        128-B segments) goes through transposed copies and the NT kernel."""
        Mtok = dy.shape[0]
        bn = 352 if cols % 352 == 0 else (384 if (cols + 383) // 384 * 384 <= (cols + 319) // 320 * 320 else 320)  # the kernel's tile width
        tiles = ((rows + 255) // 256) * ((cols + bn - 1) // bn)
        ks = max(1, min(32, 256 // tiles, Mtok // 64))  # one round of work items over the 256 CUs
        need = ks * rows * cols
        if self._slabs is None or self._slabs.numel() < need:
            self._slabs = torch.empty(need, dtype=torch.float32, device=dy.device)
            self.graphs.invalidate()  # (captured sequences hold the old buffer's address)
        rc = lib().swiftk_gemm_tn_splitk(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), self._slabs.data_ptr(), cols,
                                         rows * cols, rows, cols, Mtok, ks, _s())
        if rc == -2:  # SWIFTK_ESHAPE
            dy_t, x_t = _transpose(dy, Mtok, rows), _transpose(x, Mtok, cols)
            rc = lib().swiftk_gemm_splitk(dy_t.data_ptr(), dy_t.stride(0), x_t.data_ptr(), x_t.stride(0), self._slabs.data_ptr(),
                                          cols, rows * cols, rows, cols, Mtok, BF16, ks, _s())
        check(rc, "swiftk_gemm_tn_splitk")
        check(lib().swiftk_reduce_slabs(self._slabs.data_ptr(), cols, rows * cols, ks, out_grad.data_ptr(), out_grad.stride(0),
                                        rows, cols, int(accumulate), _s()), "swiftk_reduce_slabs")

    def _check_addresses(self):
        """Captured sequences bake in the addresses of the parameters and of their gradient buffers.  When any of them
        moved (``zero_grad(set_to_none=True)``, ``GradAllReduce.flatten_grads()`` re-pointing ``.grad`` into its flat
        buffer, ``net.to(...)``, a re-assigned parameter) every capture is stale: drop them, the next calls capture anew."""
        ptrs = tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr()) for p in self.m.parameters())
        if ptrs != getattr(self, "_ptrs", None):
            if getattr(self, "_ptrs", None) is not None:
                self.graphs.invalidate()
            self._ptrs = ptrs

    @staticmethod
    def _grad_buf(p):
        if p.grad is None:
            p.grad = _zeros(*p.shape, device=p.device)  # (accumulated into by the backward kernels)
        return p.grad

    def activation_bytes(self, B: int) -> int:
        """Bytes of saved activations of one forward pass at batch ``B`` (what a kept rollout step costs in HBM)."""
        self.refresh()
        m = self.m
        M = B * m.grid_size[0] * m.grid_size[1]
        per_token_layer = 2 * (3 * self.kd + 3 * m.dim + 2 * m.dim + 2 * self.mlp_e + self.kmlp) + 4 * 3 * m.heads
        return M * m.depth * per_token_layer

    # ------------------------------------------------------------------ forward (saves activations)
    def forward(self, srcs: Sequence[torch.Tensor], scales: Sequence[float], t, aux, want_logvar=False, slot: int = 0):
        """Forward pass that keeps the activations (``ctx``) for :meth:`backward`.  After the first call of a given
        signature the launch sequence is replayed as one HIP graph (``graphs.GraphCache``): ``out`` / ``ctx`` then are
        the capture's own tensors, overwritten by the next call with the same signature.  ``slot`` is part of the
        signature: calls whose activations must stay alive side by side (the kept rollout steps of the multistep CRPS
        loss) use different slots and so get their own buffers."""
        self.refresh()
        srcs = [s.contiguous().float() for s in srcs]
        ins = list(srcs) + [t.contiguous().float()] + ([aux.contiguous().float()] if aux is not None else [])
        key = ("fwd", tuple(tuple(s.shape) for s in srcs), tuple(float(c) for c in scales), aux is not None, bool(want_logvar), int(slot))
        n = len(srcs)
        fn = lambda *a: self._forward(list(a[:n]), list(scales), a[n], a[n + 1] if aux is not None else None, want_logvar)
        res = self.graphs.call(key, fn, ins)
        res[-1]["graph_key"] = (key, max(1, self.graphs.generation.get(key, 0)))  # eager runs before the first capture count as generation 1
        return res

    def _forward(self, srcs: Sequence[torch.Tensor], scales: Sequence[float], t, aux, want_logvar=False):
        m = self.m
        dev = srcs[0].device
        B = srcs[0].shape[0]
        d, heads, mlp = m.dim, m.heads, self.mlp_e
        gh, gw = m.grid_size
        ntok = gh * gw
        M = B * ntok
        srcs = [s.contiguous().float() for s in srcs]
        ctx = dict(B=B, M=M, srcs_ch=[s.shape[1] for s in srcs], scales=list(scales), layers=[])
        # time embedding -> latent -> modulation
        t = t.contiguous().float()
        aux_s = None
        if m.auxiliary_embed is not None and aux is not None:
            aux_s = aux.contiguous().float()
        emb = ops.timestep_embed(t, aux_s, self.freqs,
                                 None if aux_s is None else m.auxiliary_embed.weight.detach().float().contiguous(),
                                 None if aux_s is None else m.auxiliary_embed.bias.detach().float().contiguous(), d,
                                 float(m.timestep_weight))
        z1 = self._small(emb, m.latent_embed.l1.weight, m.latent_embed.l1.bias, 0)
        h1 = self._small(emb, m.latent_embed.l1.weight, m.latent_embed.l1.bias, 1)
        z2 = self._small(h1, m.latent_embed.l2.weight, m.latent_embed.l2.bias, 0)
        lat = self._small(h1, m.latent_embed.l2.weight, m.latent_embed.l2.bias, 1)
        mod = ops.linear_small(lat, self.mod_w, self.mod_b, 0)  # [B, depth*2*2d]
        ctx.update(aux=aux_s, emb=emb, z1=z1, h1=h1, z2=z2, lat=lat, mod=mod)
        logvar = None
        if want_logvar:
            logvar = self._small(lat, m.logvar_embed.weight, m.logvar_embed.bias, 0).reshape(B)
        # patch embedding
        ape = ops.patchify(srcs, scales, m.patch_size, self.kpe, _BF)
        x = torch.empty(M, d, dtype=torch.float32, device=dev)
        _gemm(ape, self.pe, x, EPI_BIAS_POS, m.patch_embed.emb.bias.detach().float().contiguous(),
              m.pos_embed.detach().reshape(ntok, d).float().contiguous(), ntok)
        # residual stream as in the inference engine (round 4): hi = bf16(x) IS the GEMM operand -- one saved activation per
        # branch -- plus an 8-bit low part updated in place; 8 instead of 12 bytes per element and launch.  Row widths the packed
        # kernel is not built for keep the fp32 stream + operand copy.
        pair = d in (1056, 1280) and ntok % 16 == 0 and not os.environ.get("SWIFTK_TRAIN_FP32_STREAM")
        if pair:
            xT, xlo = ops.split_pair(x, self.kd, 8)
        else:
            xT = ops.pad_cols(x, self.kd, _BF)
        ctx["ape"] = ape

        def norm_res(yb, g, bta, msl, hi_in):
            """x += ModulatedNorm(yb); returns the new GEMM operand."""
            hi_out = _padded(M, self.kd, d)
            if pair:
                check(lib().swiftk_modnorm_residual_pair_to(yb.data_ptr(), yb.stride(0), hi_in.data_ptr(), hi_out.data_ptr(), self.kd,
                                                            xlo.data_ptr(), d, 8, g.data_ptr(), bta.data_ptr(), msl.data_ptr(),
                                                            msl.stride(0), M, d, ntok, 1e-6, _s()), "swiftk_modnorm_residual_pair_to")
            else:
                ops.modnorm_residual(yb, x, g, bta, msl, ntok, xcopy=hi_out)
            return hi_out

        do_shift = any(m.shift_size)
        for i, (att, ff) in enumerate(m.transformer.layers):
            W = self.L[i]
            sh = tuple(m.shift_size) if (do_shift and i % 2) else (0, 0)
            qkvh = torch.empty(M, 3 * d, dtype=_BF, device=dev)
            rn = torch.empty(M, 3 * heads, dtype=torch.float32, device=dev)
            _gemm(xT, W["qkv"], qkvh, EPI_QKNORM, W["scale"], rn, pos_rows=self.hd, k=d)  # (QKNORM: head_dim rides in pos_rows)
            a = _padded(M, self.kd, d)
            ops.window_attention(qkvh.view(B, ntok, 3 * d), None, (gh, gw), heads, sh, out=a.view(B, ntok, self.kd),
                                 flags=ATTN_PRENORM)
            y1 = torch.empty(M, d, dtype=_BF, device=dev)
            _gemm(a, W["wo"], y1, k=d)
            msl1 = mod[:, (2 * i) * 2 * d:(2 * i + 1) * 2 * d]
            xT_mid = norm_res(y1, att.norm.norm.weight.detach().float(), att.norm.norm.bias.detach().float(), msl1, xT)
            h = torch.empty(M, 2 * mlp, dtype=_BF, device=dev)
            hmid = _padded(M, self.kmlp, mlp)
            if mlp % 8 == 0:  # h (kept for the backward pass) and silu(gate) * up out of one GEMM epilogue
                _gemm(xT_mid, W["w1"], h, EPI_SWIGLU_BOTH, None, hmid, pos_rows=hmid.stride(0), k=d)
            else:
                _gemm(xT_mid, W["w1"], h, k=d)
                check(lib().swiftk_swiglu_fwd(h.data_ptr(), h.stride(0), hmid.data_ptr(), hmid.stride(0), M, mlp, BF16, _s()),
                      "swiftk_swiglu_fwd")
            y2 = torch.empty(M, d, dtype=_BF, device=dev)
            _gemm(hmid, W["w2"], y2)
            msl2 = mod[:, (2 * i + 1) * 2 * d:(2 * i + 2) * 2 * d]
            xT_out = norm_res(y2, ff.norm.norm.weight.detach().float(), ff.norm.norm.bias.detach().float(), msl2, xT_mid)
            ctx["layers"].append(dict(xT_in=xT, qkvh=qkvh, rn=rn, att=a, y1=y1, xT_mid=xT_mid, h=h, hmid=hmid, y2=y2, shift=sh))
            xT = xT_out
        ctx["xT_final"] = xT
        po = m.out_channels * m.patch_size[0] * m.patch_size[1]
        po4 = self.head.shape[0]
        tok = torch.empty(M, po4, dtype=torch.float32, device=dev)
        _gemm(xT, self.head, tok, k=d)
        out = ops.unpatchify_affine(tok.view(B, ntok, po4), (B, m.out_channels, *m.image_size), m.patch_size)
        return (out, logvar, ctx) if want_logvar else (out, ctx)

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, dout: torch.Tensor, dlogvar: Optional[torch.Tensor] = None, need_input_grad: Sequence[bool] = (),
                 grads_final=None):
        """See :meth:`_backward`.  Without a ``grads_final`` hook (single process, or not the last pass of the iteration) the
        launch sequence is replayed as a HIP graph; with one it runs eagerly so the per-layer all-reduces can be started
        from Python between the layers."""
        from .dist import collectives_active
        if grads_final is not None and not collectives_active():
            grads_final = None  # no process group: nothing to reduce
        if grads_final is not None:
            return self._backward(ctx, dout, dlogvar, need_input_grad, grads_final)
        for p in self.m.parameters():  # gradient buffers must exist (fixed addresses) before anything is captured
            self._grad_buf(p)
        self._check_addresses()
        ins = [dout.contiguous().float()] + ([dlogvar.contiguous().float()] if dlogvar is not None else [])
        key = ("bwd", ctx.get("graph_key"), dlogvar is not None, tuple(bool(b) for b in need_input_grad))
        fn = lambda *a: self._backward(ctx, a[0], a[1] if dlogvar is not None else None, need_input_grad, None)
        return self.graphs.call(key, fn, ins, transient=True)  # input gradients are consumed at once by the losses

    def _backward(self, ctx, dout: torch.Tensor, dlogvar: Optional[torch.Tensor] = None, need_input_grad: Sequence[bool] = (),
                  grads_final=None):
        """Accumulate parameter gradients into .grad; return input gradients for the sources flagged in need_input_grad.

        ``grads_final`` (callable taking a list of parameters): this is the LAST backward pass of the iteration, so a
        parameter's gradient is complete once this pass has added its contribution -- the callable is told layer by
        layer, and the data-parallel wrapper starts that slice's all-reduce while the remaining layers still compute
        (what DDP's bucketed all-reduce does for the reference, trainer.py:76-84)."""
        m = self.m
        dev = dout.device
        B, M = ctx["B"], ctx["M"]
        d, heads, mlp, mlp0 = m.dim, m.heads, self.mlp_e, m.mlp_dim
        gh, gw = m.grid_size
        ntok = gh * gw
        L = lib()
        po = m.out_channels * m.patch_size[0] * m.patch_size[1]
        G = self._grad_buf
        # the ModulatedNorm backward's column-sum workspace is kept zero by its own finishing kernel; a pass that was aborted between
        # the two kernels (a failed launch, an exception in between) would leave sums behind that every later pass adds to its
        # LayerNorm / modulation gradients -- so each pass re-establishes the invariant with ONE clear up front
        if getattr(self, "_row_stats", None) is not None and self._row_stats.numel() >= 2 * B * d:
            ops.zero_acc_(self._row_stats[:2 * B * d])
        # ---- head: tok = xT_final @ Whead^T ; dtok = patchify(dout)
        dtok = ops.patchify([dout.contiguous().float()], [1.0], m.patch_size, self.kpo, _BF)  # [M, kpo] bf16, pad zero
        # feature order of the head is (c, p1, p2) while patchify emits (p1, p2, c): permute columns accordingly
        p1, p2 = m.patch_size
        C = m.out_channels
        perm = torch.arange(po, device=dev).view(p1 * p2, C).t().reshape(-1)  # head col (c*p1*p2 + pp) <- patchify col pp*C + c
        dtok_h = torch.zeros_like(dtok)
        dtok_h[:, :po] = dtok[:, perm]
        dx = torch.empty(M, d, dtype=torch.float32, device=dev)
        _gemm(dtok_h, self.head_t, dx)                                     # dgrad -> d xT_final
        # the split-K kernel wants row counts in multiples of 8: use the zero-padded kpo rows and keep the first po
        gh_pad = torch.empty(self.kpo, d, dtype=torch.float32, device=dev)
        self._wgrad(dtok_h, ctx["xT_final"], self.kpo, d, gh_pad, accumulate=False)
        G(m.head.head[0].weight).add_(gh_pad[:po])
        if grads_final is not None:
            grads_final(list(m.head.parameters()))
        dmod = _zeros(B, m.depth * 4 * d, device=dev)
        # the backward pass's temporaries live for one layer each: one set for the whole pass (their k-paddings are zeroed once
        # instead of once per layer; every kernel writes the valid columns only)
        dy2, dy1 = _padded(M, self.kd, d), _padded(M, self.kd, d)
        dh = _padded(M, max(self.kh, 2 * mlp), 2 * mlp)
        dqkv = _padded(M, self.kqkv, 3 * d)
        datt = torch.empty(M, self.kd, dtype=_BF, device=dev)
        g1i = torch.empty(2 * mlp, d, dtype=torch.float32, device=dev)
        for i in reversed(range(m.depth)):
            att, ff = m.transformer.layers[i]
            W, A = self.L[i], ctx["layers"][i]
            mod = ctx["mod"]
            # ---- feed-forward branch
            self._modnorm_bwd(A["y2"], dx, dy2, ff.norm.norm, mod[:, (2 * i + 1) * 2 * d:(2 * i + 2) * 2 * d],
                              dmod[:, (2 * i + 1) * 2 * d:(2 * i + 2) * 2 * d], M, d, ntok)
            if mlp % 8 == 0:  # d(hidden) = dy2 @ w2 and the step back through silu(gate) * up in one launch
                _gemm(dy2, W["w2_t"], dh, EPI_SWIGLU_BWD, None, A["h"], pos_rows=A["h"].stride(0), k=d)
            else:
                dhmid = torch.empty(M, mlp, dtype=_BF, device=dev)
                _gemm(dy2, W["w2_t"], dhmid, k=d)
                check(L.swiftk_swiglu_bwd(A["h"].data_ptr(), A["h"].stride(0), dhmid.data_ptr(), dhmid.stride(0), dh.data_ptr(),
                                          dh.stride(0), M, mlp, BF16, _s()), "swiftk_swiglu_bwd")
            if mlp == mlp0:
                self._wgrad(dy2, A["hmid"], d, mlp, G(ff.w2.weight))
            else:
                g2 = torch.empty(d, mlp, dtype=torch.float32, device=dev)
                self._wgrad(dy2, A["hmid"], d, mlp, g2, accumulate=False)
                G(ff.w2.weight).add_(g2[:, :mlp0])
            _gemm(dh, W["w1_t"], dx, EPI_ACCUM)                              # residual + w1 path: dx += dh @ w1
            self._wgrad(dh, A["xT_mid"], 2 * mlp, d, g1i, accumulate=False)
            G(ff.w1.weight).add_(g1i[:2 * mlp0].view(mlp0, 2, d).permute(1, 0, 2).reshape(2 * mlp0, d))  # undo the interleave
            # ---- attention branch
            self._modnorm_bwd(A["y1"], dx, dy1, att.norm.norm, mod[:, (2 * i) * 2 * d:(2 * i + 1) * 2 * d],
                              dmod[:, (2 * i) * 2 * d:(2 * i + 1) * 2 * d], M, d, ntok)
            _gemm(dy1, W["wo_t"], datt, k=d)  # N = d columns written, row stride kd
            self._wgrad(dy1, A["att"], d, d, G(att.wo.weight))
            # d(q | k | v) lands in the to_qkv data-gradient GEMM's operand buffer (row stride kqkv): the attention backward applies
            # the QK-norm backward to its accumulators on their way out (head_dim 88; elsewhere a second pass rewrites the q-hat / k-hat
            # vectors in place -- v's gradient is already final)
            sh = A["shift"]
            gscale = G(att.scale)  # [heads, 1, 1] fp32, contiguous: the kernel accumulates (atomicAdd) straight into it
            assert gscale.is_contiguous() and gscale.numel() == heads
            check(L.swiftk_window_attention_bwd_qknorm(A["qkvh"].data_ptr(), 3 * d, A["att"].data_ptr(), datt.data_ptr(), self.kd,
                                                       dqkv.data_ptr(), self.kqkv, W["scale"].data_ptr(), A["rn"].data_ptr(),
                                                       gscale.data_ptr(), B, gh, gw, heads, self.hd, sh[0], sh[1], BF16, _s()),
                  "swiftk_window_attention_bwd_qknorm")
            _gemm(dqkv, W["qkv_t"], dx, EPI_ACCUM, k=3 * d)
            self._wgrad(dqkv, A["xT_in"], 3 * d, d, G(att.to_qkv.weight))
            if grads_final is not None:  # everything of layer i except its modulation Linears (those follow in _embed_bwd)
                grads_final([p for n, p in m.transformer.layers[i].named_parameters() if "modulation" not in n])
        # ---- patch embedding: x0 = ape @ Wpe^T + b + pos
        G(m.pos_embed).view(ntok, d)  # ensure buffer exists
        dxb = torch.empty(M, self.kd, dtype=_BF, device=dev)
        check(L.swiftk_embed_bwd_sums(dx.data_ptr(), d, G(m.patch_embed.emb.bias).data_ptr(), m.pos_embed.grad.data_ptr(),
                                      dxb.data_ptr(), self.kd, M, d, ntok, _s()), "swiftk_embed_bwd_sums")
        pf = m.in_channels * p1 * p2
        gpe = torch.empty(d, self.kpe, dtype=torch.float32, device=dev)  # kpe (multiple of 8) columns; the pad ones are 0
        self._wgrad(dxb, ctx["ape"], d, self.kpe, gpe, accumulate=False)
        G(m.patch_embed.emb.weight).add_(gpe[:, :pf])
        dins: List[Optional[torch.Tensor]] = []
        if any(need_input_grad):
            dape = torch.empty(M, pf, dtype=torch.float32, device=dev)
            _gemm(dxb, self.pe_t[:pf], dape)                                 # [M, pf] grads of the patch features
            # features are (pp, c) ordered over the concatenated channels: scatter back per source
            Cin = m.in_channels
            c0 = 0
            for k, need in enumerate(need_input_grad):
                ck = ctx["srcs_ch"][k]
                if need:
                    cols = (torch.arange(p1 * p2, device=dev).view(-1, 1) * Cin + torch.arange(c0, c0 + ck, device=dev).view(1, -1))
                    # unpatchify expects (c, p1, p2) feature order
                    sub = dape[:, cols.t().reshape(-1)].contiguous()
                    gk = ops.unpatchify_affine(sub.view(B, ntok, ck * p1 * p2), (B, ck, *m.image_size), m.patch_size)
                    dins.append(gk * float(ctx["scales"][k]) if ctx["scales"][k] != 1.0 else gk)
                else:
                    dins.append(None)
                c0 += ck
        # ---- modulation / latent / time embedding (fp32, B rows)
        self._embed_bwd(ctx, dmod, dlogvar)
        return dins

    def _modnorm_bwd(self, y, g, dy, ln, mod_slice, dmod_slice, M, d, ntok):
        # the per-sample column sums of the one-kernel form live in a workspace this engine keeps ZERO between calls
        # (``swiftk_modnorm_bwd_ws0``: the finishing kernel zeroes what it read, and every backward pass clears it once up front) --
        # no clear per call.  SWIFTK_MNB_MODE picks another form for A/B runs: "clear" = a clear per call inside the library (a
        # kernel; ``swiftk_set_tuning(25, 1)`` turns the library's internal clears into the hipMemsetAsync of rounds 4-5, which as a
        # node of a replayed HIP graph writes a stale pattern: DESIGN section 11), "two" = row pass + column pass.
        mode = os.environ.get("SWIFTK_MNB_MODE", "ws0")
        need = 2 * M
        if getattr(self, "_row_stats", None) is None or self._row_stats.numel() < need:
            self._row_stats = _zeros(need, device=y.device)
            self.graphs.invalidate()  # (captured sequences hold the old buffer's address)
        L = lib()
        args = (y.data_ptr(), y.stride(0), g.data_ptr(), dy.data_ptr(), dy.stride(0), ln.weight.detach().float().data_ptr(),
                ln.bias.detach().float().data_ptr(), mod_slice.data_ptr(), mod_slice.stride(0), self._grad_buf(ln.weight).data_ptr(),
                self._grad_buf(ln.bias).data_ptr(), dmod_slice.data_ptr(), dmod_slice.stride(0))
        tail = (M, d, ntok, 1e-6, BF16, _s())
        if mode == "ws0":
            rc = L.swiftk_modnorm_bwd_ws0(*args, self._row_stats.data_ptr(), *tail)
            if rc == 0:
                return
            if rc != -2:  # anything but SWIFTK_ESHAPE (shapes the one-kernel form does not take: the two-kernel form below)
                check(rc, "swiftk_modnorm_bwd_ws0")
            if getattr(self, "_row_stats2", None) is None or self._row_stats2.numel() < need:
                self._row_stats2 = torch.empty(need, dtype=torch.float32, device=y.device)  # (row statistics: not the zero workspace)
                self.graphs.invalidate()
            check(L.swiftk_modnorm_bwd(*args, self._row_stats2.data_ptr(), *tail), "swiftk_modnorm_bwd")
            return
        if getattr(self, "_row_stats2", None) is None or self._row_stats2.numel() < need:
            self._row_stats2 = torch.empty(need, dtype=torch.float32, device=y.device)
            self.graphs.invalidate()
        if mode == "two" and not getattr(self, "_mnb_two_kernel", False):
            L.swiftk_set_tuning(16, 0)
            self._mnb_two_kernel = True
        check(L.swiftk_modnorm_bwd(*args, self._row_stats2.data_ptr(), *tail), "swiftk_modnorm_bwd")

    def _small_bwd(self, dz, x, lin, want_dx=True, w=None, dW=None, db=None):
        w = lin.weight.detach().float().contiguous() if w is None else w
        B, N = dz.shape
        K = w.shape[1]
        dx = _zeros(B, K, device=dz.device) if want_dx else None
        dW = self._grad_buf(lin.weight) if dW is None else dW
        db = (self._grad_buf(lin.bias) if lin is not None and lin.bias is not None else None) if db is None else db
        check(lib().swiftk_linear_small_bwd(dz.data_ptr(), dz.stride(0), x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                                            None if dx is None else dx.data_ptr(), K, dW.data_ptr(), dW.stride(0),
                                            None if db is None else db.data_ptr(), B, N, K, _s()), "swiftk_linear_small_bwd")
        return dx

    def _embed_bwd(self, ctx, dmod, dlogvar):
        m = self.m
        d = m.dim
        dev = dmod.device
        # all modulation Linears: the data gradient in one launch over the concatenated [4 depth d, d] matrix (streamed once); the
        # weight / bias gradients straight into each parameter's own gradient, one small launch per Linear (a concatenated
        # buffer cost two 214-MB fills and 48 ATen adds per iteration)
        B, K = dmod.shape[0], self.mod_w.shape[1]
        lat = ctx["lat"]
        dlat = _zeros(B, K, device=dev)
        check(lib().swiftk_linear_small_bwd(dmod.data_ptr(), dmod.stride(0), None, 0, self.mod_w.data_ptr(), self.mod_w.stride(0),
                                            dlat.data_ptr(), K, None, 0, None, B, dmod.shape[1], K, _s()), "swiftk_linear_small_bwd")
        r = 0
        for att, ff in m.transformer.layers:
            for mn in (att.norm, ff.norm):
                gw, gb = self._grad_buf(mn.modulation.weight), self._grad_buf(mn.modulation.bias)
                dz = dmod[:, r:r + 2 * d]
                check(lib().swiftk_linear_small_bwd(dz.data_ptr(), dmod.stride(0), lat.data_ptr(), lat.stride(0), self.mod_w.data_ptr(),
                                                    self.mod_w.stride(0), None, 0, gw.data_ptr(), gw.stride(0), gb.data_ptr(), B, 2 * d, K,
                                                    _s()), "swiftk_linear_small_bwd")
                r += 2 * d
        if dlogvar is not None and m.logvar_embed is not None:
            dl2 = self._small_bwd(dlogvar.reshape(-1, 1).contiguous().float(), ctx["lat"], m.logvar_embed)
            ops.axpby(1.0, dlat, 1.0, dl2, out=dlat)
        dz2 = torch.empty_like(dlat)
        check(lib().swiftk_silu_bwd(ctx["z2"].data_ptr(), dlat.data_ptr(), dz2.data_ptr(), dz2.numel(), _s()), "swiftk_silu_bwd")
        dh1 = self._small_bwd(dz2, ctx["h1"], m.latent_embed.l2)
        dz1 = torch.empty_like(dh1)
        check(lib().swiftk_silu_bwd(ctx["z1"].data_ptr(), dh1.data_ptr(), dz1.data_ptr(), dz1.numel(), _s()), "swiftk_silu_bwd")
        demb = self._small_bwd(dz1, ctx["emb"], m.latent_embed.l1, want_dx=(ctx["aux"] is not None))
        if ctx["aux"] is not None and m.auxiliary_embed is not None:
            xa = (ctx["aux"] * math.sqrt(m.auxiliary_dim)).contiguous()
            self._small_bwd(demb, xa, m.auxiliary_embed, want_dx=False)
