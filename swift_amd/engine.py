"""Device-side state of one SwinV2 module for one compute dtype: GEMM-ready weights, the
``swiftk_model`` descriptor and the scratch workspace, plus the call into ``swiftk_swinv2_forward``.

Weight layout in HBM (built once per parameter version, all on the device):
  * GEMM operands (``patch_embed``, ``to_qkv``, ``wo``, ``w1``, ``w2``, ``head``) in the compute
    dtype, K padded with zero columns to the 128-byte k-tile of the MFMA kernels
    (1056 -> 1088 for bf16); ``w1`` rows interleaved (gate_j, up_j) so SwiGLU fuses into the
    GEMM epilogue;
  * everything that is not GEMM-bound stays fp32: LayerNorm affine, logit scales, pos_embed,
    the time-embedding MLP and the 2*depth modulation Linears, concatenated into one
    [depth*4*d, d] matrix so all modulation vectors come from a single launch.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import torch

from . import ops
from ._lib import BF16X3, Layer, Model, SwiftkError, check, lib


class SwinEngine:
    def __init__(self, module, dtype):
        """``dtype``: torch.bfloat16 (bf16 MFMA engine), torch.float32 (exact-fp32 MFMA engine) or the string "bf16x3" --
        the fp32 engine's kernels with every GEMM computed as three bf16 products of (hi, lo)-split operands
        (``swiftk_split3``; GEMM error 4.5e-6 against fp64 where exact fp32 has 6e-7 and plain bf16 2e-3), at 2.4-2.7 x the
        fp32 GEMM rate."""
        self.module = module
        self.dtype = dtype
        self._stamp = None
        self._keep = []
        self._ws = None
        self.model: Optional[Model] = None

    # ------------------------------------------------------------------ weights
    def _param_stamp(self):
        return tuple((p.data_ptr(), p._version) for p in self.module.parameters())

    def refresh(self) -> None:
        stamp = self._param_stamp()
        if stamp == self._stamp:
            return
        m = self.module
        dev = m.pos_embed.device
        if dev.type != "cuda":
            raise SwiftkError("SwinV2 parameters must live on the GPU (module.to('cuda')); there is no CPU path")
        x3 = self.dtype == "bf16x3"
        # which GEMMs of the bf16x3 engine stay on the exact-fp32 kernel (bit 0 to_qkv, 1 wo, 2 w1, 3 w2, 4 patch embed, 5 head)
        # Default 17: the cosine logits multiply q-hat . k-hat by up to 100, and an error in the patch embedding passes through
        # every layer -- measured on Swift-B against the reference: all GEMMs split 2.8e-4, these two exact 7.9e-5 (exact
        # engine 4.0e-5); 68 against 47 sample-steps/s for the exact engine at 8 units per step.
        # bit 6 (round 4, the default with bit 4): to_qkv split too, except the head PAIRS whose logit scale
        # exp(min(scale, ln 100)) exceeds SWIFTK_X3_TAU (default 25) -- those are recomputed on the exact kernel (the split
        # product's 4.5e-6 reaches the softmax multiplied by the scale: 1.1e-4 at 25, 4.5e-4 at the clamp's 100).
        exact_mask = int(os.environ.get("SWIFTK_X3_EXACT", "80"))  # travels in the model descriptor (mo.x3_exact), per engine
        tau_max = float(os.environ.get("SWIFTK_X3_TAU", "25"))
        adaptive = (x3 and bool(exact_mask & 64) and not (exact_mask & 1) and m.heads % 2 == 0
                    and (m.dim // m.heads) in (80, 88, 96))
        if x3 and (exact_mask & 64) and not adaptive:
            # the hot-pair recompute exists for head_dim 80 / 88 / 96 and an even head count only: any other shape keeps to_qkv on the exact
            # kernel (bit 0, the round-3 default) instead of running it fully split with no recompute (2.8e-4 on Swift-B)
            exact_mask = (exact_mask & ~64) | 1
        dt = torch.float32 if x3 else self.dtype  # activations, k-paddings and every non-GEMM kernel
        d, heads, depth, mlp = m.dim, m.heads, m.depth, m.mlp_dim
        p1, p2 = m.patch_size
        # int(8/3 * dim) is odd for some widths (1280 -> 3413): zero (gate, up) row pairs up to a multiple of 8 make w1's N a
        # multiple of 16, which the persistent GEMM's whole-tile path wants of a SwiGLU output (16-byte row chunks of N / 2 bf16
        # columns; N = 6828 sent the 468 M variant's w1 to the one-tile-per-workgroup kernel: 0.38 of peak against 0.50); the
        # extra SwiGLU columns are silu(0) * 0 = 0 and land in w2's zero K padding
        mlp_e = (mlp + 7) // 8 * 8
        kd, kmlp, kpe = ops.k_pad(dt, d), ops.k_pad(dt, mlp_e), ops.k_pad(dt, m.in_channels * p1 * p2)
        keep = []

        def f32(t):
            t = t.detach().contiguous().float()
            keep.append(t)
            return t.data_ptr()

        def gemm_w(t, k, exact=False):
            if x3 and not exact:  # [hi | hi | lo] blocks over the valid columns (rounded up to 4), row stride k_pad(bf16, 3 K)
                t = ops.split3(t.detach(), 1, cols=(t.shape[1] + 3) // 4 * 4)
            else:
                t = ops.pad_cols(t.detach(), k, dt)
            keep.append(t)
            return t.data_ptr()

        layers = (Layer * depth)()
        mods_w, mods_b = [], []
        for i, (att, ff) in enumerate(m.transformer.layers):
            w1 = ff.w1.weight.detach()
            w1i = w1.view(2, mlp, d).permute(1, 0, 2).reshape(2 * mlp, d)  # rows: gate_0, up_0, gate_1, up_1, ...
            if mlp_e != mlp:
                w1i = torch.cat([w1i, w1i.new_zeros(2 * (mlp_e - mlp), d)], 0)
            layers[i].qkv_w = gemm_w(att.to_qkv.weight, kd, exact=bool(exact_mask & 1))
            if adaptive:
                tau = torch.exp(torch.clamp(att.scale.detach().reshape(-1).float(), max=math.log(100.0))).cpu()
                hot = 0
                for pp in range(heads // 2):
                    if float(tau[2 * pp:2 * pp + 2].max()) > tau_max:
                        hot |= 1 << pp
                if heads <= 16:  # the hot heads themselves in bits 16..31 (include/swiftk.h): each is recomputed alone
                    for h in range(heads):
                        if float(tau[h]) > tau_max:
                            hot |= 1 << (16 + h)
                    if hot & (1 << 31):
                        hot -= 1 << 32  # (the field is a signed 32-bit integer)
                layers[i].qk_exact_pairs = hot
                layers[i].qkv_w_f32 = gemm_w(att.to_qkv.weight, kd, exact=True) if hot & 0xffff else None
            layers[i].wo_w = gemm_w(att.wo.weight, kd, exact=bool(exact_mask & 2))
            layers[i].w1_w = gemm_w(w1i, kd, exact=bool(exact_mask & 4))
            layers[i].w2_w = gemm_w(ff.w2.weight, kmlp, exact=bool(exact_mask & 8))
            layers[i].scale = f32(att.scale.reshape(-1))
            layers[i].ln1_g, layers[i].ln1_b = f32(att.norm.norm.weight), f32(att.norm.norm.bias)
            layers[i].ln2_g, layers[i].ln2_b = f32(ff.norm.norm.weight), f32(ff.norm.norm.bias)
            mods_w += [att.norm.modulation.weight.detach(), ff.norm.modulation.weight.detach()]
            mods_b += [att.norm.modulation.bias.detach(), ff.norm.modulation.bias.detach()]

        half = d // 2
        freqs = torch.exp(-math.log(10_000) * torch.arange(half, dtype=torch.float32, device=dev) / half)  # swinv2.py:48-50

        mo = Model()
        mo.dtype = BF16X3 if x3 else ops.dtype_code(dt)
        mo.H, mo.W = m.image_size
        mo.p1, mo.p2 = p1, p2
        mo.in_ch, mo.out_ch = m.in_channels, m.out_channels
        mo.depth, mo.dim, mo.heads, mo.mlp = depth, d, heads, mlp_e
        mo.wh, mo.ww = m.window_size
        mo.sh, mo.sw = m.shift_size
        mo.aux_dim = m.auxiliary_dim
        mo.has_logvar = int(m.logvar_embed is not None)
        mo.x3_exact = exact_mask if x3 else 0
        mo.timestep_weight = float(m.timestep_weight)
        mo.kd, mo.kmlp, mo.kpe = kd, kmlp, kpe
        mo.pe_w = gemm_w(m.patch_embed.emb.weight, kpe, exact=bool(exact_mask & 16))
        mo.pe_b = f32(m.patch_embed.emb.bias)
        mo.pos = f32(m.pos_embed.reshape(-1, d))
        mo.freqs = f32(freqs)
        if m.auxiliary_embed is not None:
            mo.aux_w, mo.aux_b = f32(m.auxiliary_embed.weight), f32(m.auxiliary_embed.bias)
        mo.l1_w, mo.l1_b = f32(m.latent_embed.l1.weight), f32(m.latent_embed.l1.bias)
        mo.l2_w, mo.l2_b = f32(m.latent_embed.l2.weight), f32(m.latent_embed.l2.bias)
        mo.mod_w, mo.mod_b = f32(torch.cat(mods_w, 0)), f32(torch.cat(mods_b, 0))
        if m.logvar_embed is not None:
            mo.logvar_w, mo.logvar_b = f32(m.logvar_embed.weight), f32(m.logvar_embed.bias)
        hw = m.head.head[0].weight.detach()
        if hw.shape[0] % 4:  # GEMM N granularity: zero rows (1x1 patches: 69 -> 72 output columns, the extra ones unused)
            hw = torch.cat([hw, hw.new_zeros(4 - hw.shape[0] % 4, hw.shape[1])], 0)
        mo.head_w = gemm_w(hw, kd, exact=bool(exact_mask & 32))
        mo.layers_host = C.cast(layers, C.POINTER(Layer))
        keep.append(layers)
        self.model, self._keep, self._stamp = mo, keep, stamp

    # ------------------------------------------------------------------ forward
    def workspace(self, B: int) -> torch.Tensor:
        need = int(lib().swiftk_workspace_bytes(C.byref(self.model), B))
        if need <= 0:
            raise SwiftkError("model configuration not supported by the gfx950 kernels "
                              "(need 16x16 windows, grid divisible by 16, dim % 4 == 0)")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.module.pos_embed.device)
        return self._ws

    def forward(self, srcs: Sequence[torch.Tensor], scales: Sequence[float], t: torch.Tensor,
                aux: Optional[torch.Tensor], xt: Optional[torch.Tensor] = None, alpha: Optional[torch.Tensor] = None,
                beta: Optional[torch.Tensor] = None, want_logvar: bool = False):
        """out = alpha*xt + beta*SwinV2(cat_k srcs[k]*scales[k], t, aux)   (all fp32 NCHW on the device)."""
        self.refresh()
        m = self.module
        B = srcs[0].shape[0]
        srcs = [s.contiguous().float() for s in srcs]
        for s in srcs:
            if not s.is_cuda:
                raise SwiftkError("inputs must be device tensors; there is no CPU path")
        t = t.contiguous().float()
        aux = None if aux is None else aux.contiguous().float()
        ws = self.workspace(B)
        out = torch.empty(B, m.out_channels, *m.image_size, dtype=torch.float32, device=t.device)
        logvar = torch.empty(B, dtype=torch.float32, device=t.device) if want_logvar else None
        ps = [(s.data_ptr(), s.shape[1], float(c)) for s, c in zip(srcs, scales)] + [(None, 0, 1.0)] * (3 - len(srcs))
        keep = (srcs, t, aux, xt, alpha, beta)  # noqa: F841  (alive until the launch sequence is enqueued)
        rc = lib().swiftk_swinv2_forward(
            C.byref(self.model), ps[0][0], ps[0][1], ps[0][2], ps[1][0], ps[1][1], ps[1][2], ps[2][0], ps[2][1], ps[2][2],
            t.data_ptr(), None if aux is None else aux.data_ptr(), None if xt is None else xt.data_ptr(),
            None if alpha is None else alpha.data_ptr(), None if beta is None else beta.data_ptr(), out.data_ptr(),
            None if logvar is None else logvar.data_ptr(), B, ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream().cuda_stream)
        check(rc, "swiftk_swinv2_forward")
        return (out, logvar) if want_logvar else out
