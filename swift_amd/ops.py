"""Tensor-level wrappers over the C ABI (``include/swiftk.h``).

PyTorch is plumbing here: it owns device memory and the stream; every wrapper
passes raw ``data_ptr()``s to the HIP kernels and returns immediately (the
launch is asynchronous on torch's current stream).  CPU tensors are rejected --
there is no fallback implementation.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (ATTN_NO_PIPE, ATTN_PRENORM, ATTN_TILED, BF16, EPI_ACCUM, EPI_BIAS_POS, EPI_NONE, EPI_QKNORM, EPI_SWIGLU, F32, SwiftkError,
                   check, lib)

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise SwiftkError("swift_amd kernels need device (HIP) tensors; there is no CPU fallback")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise SwiftkError(f"unsupported dtype {dt}") from None


def k_pad(dtype: torch.dtype, k: int) -> int:
    return int(lib().swiftk_gemm_k_pad(dtype_code(dtype), k))


def pad_cols(w: torch.Tensor, k: int, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[rows, cols] fp32 -> [rows, k] ``dtype`` with zero padding (GEMM operand layout), on the device.
    ``out``: refill an existing operand buffer (same address for captured launch sequences)."""
    _dev(w)
    w = w.contiguous().float()
    if out is None:
        out = torch.empty(w.shape[0], k, dtype=dtype, device=w.device)
    assert out.shape == (w.shape[0], k) and out.dtype == dtype and out.is_contiguous()
    check(lib().swiftk_cast_pad(w.data_ptr(), w.shape[1], out.data_ptr(), k, w.shape[0], w.shape[1], dtype_code(dtype),
                                _stream()), "swiftk_cast_pad")
    return out


def cast_pad_t(w: torch.Tensor, out: torch.Tensor, out_t: torch.Tensor, interleave: int = 0) -> None:
    """bf16 operand copies of an fp32 weight in one pass: ``out`` [rows, >=cols] = w (w1: rows (gate, up)-interleaved when
    ``interleave`` = mlp), ``out_t`` [cols, >=rows] = its transpose; row paddings zeroed."""
    _dev(w, out, out_t)
    assert w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1 and out.dtype == out_t.dtype == torch.bfloat16
    assert out.shape[0] == w.shape[0] and out_t.shape[0] == w.shape[1] and out.stride(1) == 1 and out_t.stride(1) == 1
    check(lib().swiftk_cast_pad_t(w.data_ptr(), w.stride(0), w.shape[0], w.shape[1], out.data_ptr(), out.stride(0), out_t.data_ptr(),
                                  out_t.stride(0), int(interleave), _stream()), "swiftk_cast_pad_t")


def split3(w: torch.Tensor, order: int, cols: Optional[int] = None) -> torch.Tensor:
    """fp32 [rows, c] -> bf16 [rows, k_pad(bf16, 3 * cols)]: the three-block (hi / lo) operand of the bf16x3 engine
    (``swiftk_split3``; order 0 = activations [hi | lo | hi], 1 = weights [hi | hi | lo]).  ``cols`` >= c: zero columns are
    appended first (valid K rounded up to the split's granularity of 4)."""
    _dev(w)
    w = w.contiguous().float()
    cols = cols or w.shape[1]
    if cols != w.shape[1]:
        w = torch.nn.functional.pad(w, (0, cols - w.shape[1]))
    out = torch.empty(w.shape[0], k_pad(torch.bfloat16, 3 * cols), dtype=torch.bfloat16, device=w.device)
    check(lib().swiftk_split3(w.data_ptr(), w.shape[1], out.data_ptr(), out.shape[1], w.shape[0], cols, order, _stream()),
          "swiftk_split3")
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, *, n_out: Optional[int] = None,
         out_dtype: Optional[torch.dtype] = None, epilogue: int = EPI_NONE, bias: Optional[torch.Tensor] = None,
         pos: Optional[torch.Tensor] = None, head_dim: int = 0) -> torch.Tensor:
    """out[M, N] = epilogue(a[M, K] @ w[N, K]^T); a, w row-major 2-D (row stride may exceed K).
    ``head_dim``: EPI_QKNORM only (80 / 88 / 96; 0 = 88)."""
    _dev(a, w, out, bias, pos)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    odt = out_dtype or a.dtype
    ncol = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out or ncol, dtype=odt, device=a.device)
    check(lib().swiftk_gemm(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                            dtype_code(a.dtype), dtype_code(out.dtype), epilogue, _ptr(bias), _ptr(pos),
                            (head_dim if epilogue == EPI_QKNORM else 0) if pos is None else pos.shape[0], _stream()), "swiftk_gemm")
    return out


_chunk_scratch = {}


def gemm_chunked(a: torch.Tensor, w: torch.Tensor, *, chunk_k: int = 256, out_dtype: Optional[torch.dtype] = None,
                 epilogue: int = EPI_NONE, bias: Optional[torch.Tensor] = None, pos: Optional[torch.Tensor] = None,
                 head_dim: int = 0) -> torch.Tensor:
    """``gemm`` for fp32 operands with two-level accumulation (``swiftk_gemm_chunked``): MFMA chains of ``chunk_k``, partial
    sums met in fp32 through a scratch slab -- the exact-fp32 engine's product."""
    _dev(a, w, bias, pos)
    M, K = a.shape
    N = w.shape[0]
    assert a.dtype == w.dtype == torch.float32 and w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    need = int(lib().swiftk_gemm_chunk_scratch_bytes())
    # (the kernel parks partial accumulators in the slab by blockIdx / wave: one slab per stream, so that two calls in flight on
    # different streams never share it)
    key = (a.device, _stream())
    scr = _chunk_scratch.get(key)
    if scr is None or scr.numel() < need:
        scr = _chunk_scratch[key] = torch.empty(need, dtype=torch.uint8, device=a.device)
    out = torch.empty(M, N // 2 if epilogue == EPI_SWIGLU else N, dtype=out_dtype or torch.float32, device=a.device)
    check(lib().swiftk_gemm_chunked(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                                    dtype_code(a.dtype), dtype_code(out.dtype), epilogue, _ptr(bias), _ptr(pos),
                                    (head_dim if epilogue == EPI_QKNORM else 0) if pos is None else pos.shape[0], int(chunk_k),
                                    scr.data_ptr(), scr.numel(), _stream()), "swiftk_gemm_chunked")
    return out


def gemm_qkv_tiled(a: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, B: int, grid: Tuple[int, int], heads: int,
                   shift: Tuple[int, int] = (0, 0), out: Optional[torch.Tensor] = None, k: Optional[int] = None,
                   head_dim: int = 88) -> torch.Tensor:
    """to_qkv with the QK-norm epilogue, stored window-tiled: out[B, windows, heads, 3, 256, head_dim] (bf16).

    ``k``: valid K when it ends half-way into the last k-tile of the (padded) operand rows.
    """
    _dev(a, w, scale, out)
    gh, gw = grid
    assert a.shape[0] == B * gh * gw and w.shape[0] == 3 * heads * head_dim and a.dtype == torch.bfloat16
    if out is None:
        out = torch.empty(B, (gh // 16) * (gw // 16), heads, 3, 256, head_dim, dtype=torch.bfloat16, device=a.device)
    check(lib().swiftk_gemm_qkv_tiled(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(),
                                      k or a.shape[1], scale.contiguous().data_ptr(), B, gh, gw, heads, head_dim, shift[0],
                                      shift[1], _stream()), "swiftk_gemm_qkv_tiled")
    return out


def window_attention_tiled(qkv_tiled: torch.Tensor, scale: Optional[torch.Tensor], grid: Tuple[int, int], heads: int,
                           shift: Tuple[int, int] = (0, 0), out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention over ``gemm_qkv_tiled`` output -> out [B, gh*gw, heads*head_dim] (token order, un-rolled)."""
    _dev(qkv_tiled, scale, out)
    B, hd = qkv_tiled.shape[0], qkv_tiled.shape[-1]
    gh, gw = grid
    if out is None:
        out = torch.empty(B, gh * gw, heads * hd, dtype=torch.bfloat16, device=qkv_tiled.device)
    check(lib().swiftk_window_attention(qkv_tiled.data_ptr(), 3 * heads * hd, out.data_ptr(), out.stride(1),
                                        None if scale is None else scale.contiguous().data_ptr(), B, gh, gw, heads, hd,
                                        shift[0], shift[1], BF16, ATTN_PRENORM | ATTN_TILED, _stream()),
          "swiftk_window_attention")
    return out


def qkv_attention_fused(a: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, B: int, grid: Tuple[int, int], heads: int,
                        shift: Tuple[int, int] = (0, 0), out: Optional[torch.Tensor] = None, k: Optional[int] = None,
                        head_dim: int = 88) -> torch.Tensor:
    """to_qkv + cosine norm + shifted-window attention in one kernel: a [B*gh*gw, >=K] bf16, w [3*heads*hd, >=K] bf16 ->
    out [B, gh*gw, heads*hd] bf16 (token order).  ``k``: valid K (may end half-way into the last k-tile of the padded rows)."""
    _dev(a, w, scale, out)
    gh, gw = grid
    assert a.shape[0] == B * gh * gw and w.shape[0] == 3 * heads * head_dim and a.dtype == w.dtype == torch.bfloat16
    if out is None:
        out = torch.empty(B, gh * gw, heads * head_dim, dtype=torch.bfloat16, device=a.device)
    check(lib().swiftk_qkv_attention_fused(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), scale.contiguous().data_ptr(),
                                           out.data_ptr(), out.stride(-2), k or a.shape[1], B, gh, gw, heads, head_dim, shift[0],
                                           shift[1], _stream()), "swiftk_qkv_attention_fused")
    return out


def window_attention(qkv: torch.Tensor, scale: Optional[torch.Tensor], grid: Tuple[int, int], heads: int,
                     shift: Tuple[int, int] = (0, 0), out: Optional[torch.Tensor] = None, flags: int = 0) -> torch.Tensor:
    """qkv [B, gh*gw, >=3*heads*hd] -> out [B, gh*gw, heads*hd] (token order, un-rolled).

    ``flags`` = ``_lib.ATTN_PRENORM`` when q/k were normalised by the ``EPI_QKNORM`` GEMM epilogue.
    """
    _dev(qkv, scale, out)
    B, n, _ = qkv.shape
    gh, gw = grid
    assert n == gh * gw and qkv.stride(2) == 1
    hd = qkv.shape[2] // (3 * heads)
    if out is None:
        out = torch.empty(B, n, heads * hd, dtype=qkv.dtype, device=qkv.device)
    check(lib().swiftk_window_attention(qkv.data_ptr(), qkv.stride(1), out.data_ptr(), out.stride(1),
                                        None if scale is None else scale.contiguous().data_ptr(), B, gh, gw, heads, hd,
                                        shift[0], shift[1], dtype_code(qkv.dtype), flags, _stream()),
          "swiftk_window_attention")
    return out


def modnorm_residual(y: torch.Tensor, x: torch.Tensor, gamma, beta, mod: torch.Tensor, rows_per_sample: int,
                     xcopy: Optional[torch.Tensor] = None, eps: float = 1e-6) -> None:
    """x[M,d] (fp32, in place) += LN(y) * (1 + mod[:, :d]) + mod[:, d:2d]; optional operand copy of the new x."""
    _dev(y, x, gamma, beta, mod, xcopy)
    M, d = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32 and mod.stride(1) == 1
    check(lib().swiftk_modnorm_residual(y.data_ptr(), y.stride(0), x.data_ptr(), _ptr(xcopy),
                                        0 if xcopy is None else xcopy.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                        mod.data_ptr(), mod.stride(0), M, d, rows_per_sample, eps, dtype_code(y.dtype),
                                        _stream()), "swiftk_modnorm_residual")


def patchify(srcs: Sequence[torch.Tensor], scales: Sequence[float], patch: Tuple[int, int], lda: int,
             dtype: torch.dtype) -> torch.Tensor:
    """Concat (channel dim) + patchify of up to three NCHW fp32 tensors -> [B*gh*gw, lda] ``dtype``."""
    _dev(*srcs)
    assert 1 <= len(srcs) <= 3
    B, _, H, W = srcs[0].shape
    srcs = [s.contiguous() for s in srcs]
    ps = [(s.data_ptr(), s.shape[1], float(c)) for s, c in zip(srcs, scales)] + [(None, 0, 1.0)] * (3 - len(srcs))
    out = torch.empty(B * (H // patch[0]) * (W // patch[1]), lda, dtype=dtype, device=srcs[0].device)
    check(lib().swiftk_patchify(ps[0][0], ps[0][1], ps[0][2], ps[1][0], ps[1][1], ps[1][2], ps[2][0], ps[2][1], ps[2][2],
                                out.data_ptr(), lda, B, H, W, patch[0], patch[1], dtype_code(dtype), _stream()),
          "swiftk_patchify")
    return out


def unpatchify_affine(tok: torch.Tensor, out_shape, patch, xt=None, alpha=None, beta=None) -> torch.Tensor:
    _dev(tok, xt, alpha, beta)
    B, C, H, W = out_shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=tok.device)
    check(lib().swiftk_unpatchify_affine(tok.data_ptr(), tok.stride(-2), _ptr(xt), _ptr(alpha), _ptr(beta), out.data_ptr(),
                                         B, C, H, W, patch[0], patch[1], _stream()), "swiftk_unpatchify_affine")
    return out


def timestep_embed(t, aux, freqs, aux_w, aux_b, d: int, timestep_weight: float = 1.0) -> torch.Tensor:
    _dev(t, aux, freqs, aux_w, aux_b)
    B = t.shape[0]
    out = torch.empty(B, d, dtype=torch.float32, device=t.device)
    ad = 0 if aux is None else aux.shape[1]
    check(lib().swiftk_timestep_embed(t.data_ptr(), _ptr(aux), freqs.data_ptr(), _ptr(aux_w), _ptr(aux_b), out.data_ptr(), B,
                                      d, ad, timestep_weight, _stream()), "swiftk_timestep_embed")
    return out


def linear_small(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0) -> torch.Tensor:
    _dev(x, w, bias)
    B, K = x.shape
    N = w.shape[0]
    out = torch.empty(B, N, dtype=torch.float32, device=x.device)
    check(lib().swiftk_linear_small(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), _ptr(bias), out.data_ptr(), N, B, N,
                                    K, act, _stream()), "swiftk_linear_small")
    return out


def rollout_update(xstd: torch.Tensor, y: torch.Tensor, mx, sx, st, phys: Optional[torch.Tensor] = None) -> None:
    """In place: phys = xstd*sx + mx + y*st ; xstd = (phys - mx)/sx  (per channel).  ``st=None``: the non-residual form of
    generate.py:132-136, phys = y*sx + mx ; xstd = y."""
    _dev(xstd, y, mx, sx, phys, *(() if st is None else (st,)))
    B, Cc, H, W = xstd.shape
    assert xstd.is_contiguous() and y.is_contiguous() and (phys is None or phys.is_contiguous())
    check(lib().swiftk_rollout_update(xstd.data_ptr(), y.data_ptr(), _ptr(phys), mx.data_ptr(), sx.data_ptr(), _ptr(st),
                                      B, Cc, H * W, _stream()), "swiftk_rollout_update")


def unit_noise(out: torch.Tensor, seeds: torch.Tensor, step=0, step_dev: Optional[torch.Tensor] = None, raw: bool = False) -> torch.Tensor:
    """out[b] ~ N(0, 1), a pure function of (seeds[b], step, element): the counter-based latent stream of production rollouts
    (Philox4x32-10 + Box-Muller, ``swiftk_unit_noise``; replaces the per-member torch generator of generate.py:83 /
    factory.py:52-56).  ``seeds`` int64 [B] on the device; the lead step is ``step`` plus ``*step_dev`` (device int64 scalar)
    when given -- a captured step reads it from memory.  ``raw``: the generator's 32-bit words instead (tests)."""
    _dev(out, seeds, step_dev)
    assert out.is_contiguous() and out.dtype == torch.float32 and seeds.dtype == torch.int64 and seeds.numel() == out.shape[0]
    assert step_dev is None or step_dev.dtype == torch.int64
    check(lib().swiftk_unit_noise(out.data_ptr(), seeds.data_ptr(), _ptr(step_dev), int(step), out.shape[0],
                                  out[0].numel(), 1 if raw else 0, _stream()), "swiftk_unit_noise")
    return out


def counter_add(counter: torch.Tensor, value: int = 1) -> None:
    """counter (device int64 scalar) += value on the current stream."""
    _dev(counter)
    assert counter.dtype == torch.int64 and counter.numel() == 1
    check(lib().swiftk_counter_add(counter.data_ptr(), int(value), _stream()), "swiftk_counter_add")


def split_pair(x: torch.Tensor, ldh: int, lo_bits: int = 8):
    """fp32 [M, d] -> (hi bf16 [M, ldh] with zeroed pad columns, lo [M, d] bf16 or uint8): the bf16 engine's residual stream."""
    _dev(x)
    assert x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 2 and lo_bits in (8, 16)
    M, d = x.shape
    hi = torch.empty(M, ldh, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(M, d, dtype=torch.bfloat16 if lo_bits == 16 else torch.uint8, device=x.device)
    check(lib().swiftk_split_pair(x.data_ptr(), d, hi.data_ptr(), ldh, lo.data_ptr(), d, lo_bits, M, d, _stream()), "swiftk_split_pair")
    return hi, lo


def pair_value(hi: torch.Tensor, lo: torch.Tensor, d: int) -> torch.Tensor:
    """The fp32 value a pair stands for (tests, diagnostics): hi + lo, or hi + (byte - 128) * ulp(hi) / 256 for the 8-bit low
    part (ulp(hi) / 256 = 2^(E - 142) for hi's biased exponent E)."""
    h = hi[:, :d].float()
    if lo.dtype == torch.bfloat16:
        return h + lo.float()
    E = (hi[:, :d].contiguous().view(torch.int16).to(torch.int32) >> 7) & 0xFF
    return h + torch.ldexp(lo.float() - 128.0, E - 142)


def modnorm_residual_pair(y: torch.Tensor, x_hi: torch.Tensor, x_lo: torch.Tensor, gamma, beta, mod: torch.Tensor,
                          rows_per_sample: int, d: int, eps: float = 1e-6) -> None:
    """In place on the pair (x_hi, x_lo): x += LayerNorm(y) * (1 + scale_b) + shift_b (swinv2.py:83-86, 211-212)."""
    _dev(y, x_hi, x_lo, gamma, beta, mod)
    assert y.dtype == x_hi.dtype == torch.bfloat16 and x_lo.dtype in (torch.bfloat16, torch.uint8) and mod.dtype == torch.float32
    M = y.shape[0]
    check(lib().swiftk_modnorm_residual_pair(y.data_ptr(), y.stride(0), x_hi.data_ptr(), x_hi.stride(0), x_lo.data_ptr(),
                                             x_lo.stride(0), 16 if x_lo.dtype == torch.bfloat16 else 8, gamma.data_ptr(),
                                             beta.data_ptr(), mod.data_ptr(), mod.stride(0), M, d, rows_per_sample, float(eps),
                                             _stream()), "swiftk_modnorm_residual_pair")


def gemm_modnorm_residual_pair(a: torch.Tensor, w: torch.Tensor, x_hi: torch.Tensor, x_lo: torch.Tensor, gamma, beta,
                               mod: torch.Tensor, rows_per_sample: int, d: int, *, k: Optional[int] = None,
                               rows_per_workgroup: int = 32, eps: float = 1e-6) -> None:
    """``modnorm_residual_pair(bf16(a @ w^T), ...)`` as one kernel over complete rows (small batches; 8-bit low part):
    a [M, >=K] bf16, w [d, >=K] bf16; ``k`` = valid K (a multiple of 32)."""
    _dev(a, w, x_hi, x_lo, gamma, beta, mod)
    assert a.dtype == w.dtype == x_hi.dtype == torch.bfloat16 and x_lo.dtype == torch.uint8 and mod.dtype == torch.float32
    assert w.shape[0] == d
    check(lib().swiftk_gemm_modnorm_residual_pair(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), k or a.shape[1],
                                                  x_hi.data_ptr(), x_hi.stride(0), x_lo.data_ptr(), x_lo.stride(0), gamma.data_ptr(),
                                                  beta.data_ptr(), mod.data_ptr(), mod.stride(0), a.shape[0], d, rows_per_sample,
                                                  float(eps), rows_per_workgroup, _stream()), "swiftk_gemm_modnorm_residual_pair")


def modnorm_residual_pair_slabs(y_slabs: torch.Tensor, x_hi: torch.Tensor, x_lo: torch.Tensor, gamma, beta, mod: torch.Tensor,
                                rows_per_sample: int, d: int, eps: float = 1e-6) -> None:
    """``modnorm_residual_pair`` with the branch output as the sum of the two fp32 slabs y_slabs[0] + y_slabs[1] ([2, M, d]):
    what a split-K wo / w2 leaves at one unit per step (bf16 slabs: ``swiftk_gemm_splitk_bf16``)."""
    _dev(y_slabs, x_hi, x_lo, gamma, beta, mod)
    assert y_slabs.dtype in (torch.float32, torch.bfloat16) and y_slabs.dim() == 3 and y_slabs.shape[0] == 2 and y_slabs.is_contiguous()
    M = y_slabs.shape[1]
    fn = lib().swiftk_modnorm_residual_pair_slabs if y_slabs.dtype == torch.float32 else lib().swiftk_modnorm_residual_pair_slabs_bf16
    check(fn(y_slabs.data_ptr(), y_slabs.stride(1), y_slabs.stride(0), x_hi.data_ptr(),
                                                   x_hi.stride(0), x_lo.data_ptr(), x_lo.stride(0),
                                                   16 if x_lo.dtype == torch.bfloat16 else 8, gamma.data_ptr(), beta.data_ptr(),
                                                   mod.data_ptr(), mod.stride(0), M, d, rows_per_sample, float(eps), _stream()),
          "swiftk_modnorm_residual_pair_slabs")


def axpby(a: float, x: torch.Tensor, b: float, y: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a*x + b*y on fp32 device tensors of equal shape (out may alias x or y)."""
    _dev(x, y, out)
    assert x.is_contiguous() and y.is_contiguous() and x.shape == y.shape and x.dtype == y.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    check(lib().swiftk_axpby(out.data_ptr(), float(a), x.data_ptr(), float(b), y.data_ptr(), x.numel(), _stream()),
          "swiftk_axpby")
    return out


def zero_acc_(t: torch.Tensor) -> torch.Tensor:
    """Zero a contiguous fp32 device tensor that a kernel is about to ACCUMULATE into (atomics / read-modify-write) with the
    library's own fill kernel on the current stream (``swiftk_zero_f32``).  Every zero-then-accumulate site of the training
    path goes through here or ``zeros_acc``: what clears the target is then a known, ordinary kernel launch -- never a device
    memset, which, captured into a HIP graph and replayed on the null stream, writes a stale pattern under this PyTorch's HIP
    runtime (round 5's gradient overflow: DESIGN section 11, ``tools/memset_graph_repro.hip``)."""
    _dev(t)
    assert t.dtype == torch.float32 and t.is_contiguous()
    check(lib().swiftk_zero_f32(t.data_ptr(), t.numel(), _stream()), "swiftk_zero_f32")
    return t


def zeros_acc(*shape, device) -> torch.Tensor:
    """``zero_acc_`` of a fresh fp32 tensor."""
    return zero_acc_(torch.empty(*shape, dtype=torch.float32, device=device))


def unit_checksum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp64 sum of every unit of x [B, ...] fp32 in a fixed order (bit-stable across ranks and batch slots) -> [B] fp64."""
    _dev(x, out)
    assert x.is_contiguous() and x.dtype == torch.float32
    B = x.shape[0]
    if out is None:
        out = torch.empty(B, dtype=torch.float64, device=x.device)
    scratch = torch.empty(32 * B, dtype=torch.float64, device=x.device)
    check(lib().swiftk_unit_checksum(x.data_ptr(), out.data_ptr(), scratch.data_ptr(), B, x.numel() // B, _stream()),
          "swiftk_unit_checksum")
    return out
