"""ctypes binding of ``csrc/libswiftk.so`` (the C ABI declared in ``include/swiftk.h``).

There is deliberately no fallback: if the shared library is missing or a kernel
rejects its arguments, the caller gets an exception -- the product path never
silently computes on the CPU or through ATen.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SWIFTK_LIB: A/B builds of the same ABI (csrc/Makefile `variant` target); the product always loads csrc/libswiftk.so
LIB_PATH = os.environ.get("SWIFTK_LIB") or os.path.join(_HERE, "csrc", "libswiftk.so")

F32, BF16 = 0, 1
BF16X3 = 2  # swiftk_model.dtype only: fp32 activations, every GEMM as three bf16 products (include/swiftk.h)
EPI_NONE, EPI_BIAS_POS, EPI_SWIGLU, EPI_QKNORM, EPI_ACCUM, EPI_SWIGLU_BOTH, EPI_SWIGLU_BWD = 0, 1, 2, 3, 5, 6, 7
EPI_QKNORM_JVP, EPI_SWIGLU_JVP = 8, 9  # swiftk_gemm_jvp only
EPI_SWIGLU_SPLIT3 = 10
ATTN_PRENORM, ATTN_NO_PIPE, ATTN_TILED = 1, 2, 4
PROF_ATTENTION = 100

_ERR = {-1: "SWIFTK_EINVAL (bad argument)", -2: "SWIFTK_ESHAPE (unsupported shape)",
        -3: "SWIFTK_EALIGN (misaligned pointer / leading dimension)", -4: "SWIFTK_EWORKSPACE (workspace too small)"}


class SwiftkError(RuntimeError):
    pass


class Layer(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in
                 ("qkv_w", "wo_w", "w1_w", "w2_w", "scale", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "qkv_w_f32")]
                + [("qk_exact_pairs", C.c_int32)])


OPT_MAX_GROUPS = 8


class OptChunk(C.Structure):
    _fields_ = [("p", C.c_void_p), ("ema", C.c_void_p), ("flat_off", C.c_int64), ("n", C.c_int32), ("group", C.c_int32)]


class OptHyper(C.Structure):
    _fields_ = [("lr", C.c_float * OPT_MAX_GROUPS), ("weight_decay", C.c_float * OPT_MAX_GROUPS),
                ("step_size", C.c_float * OPT_MAX_GROUPS), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bias2_sqrt", C.c_float), ("ema_beta", C.c_float), ("decoupled", C.c_int32)]


class Model(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("dtype", "H", "W", "p1", "p2", "in_ch", "out_ch", "depth", "dim", "heads", "mlp",
                                  "wh", "ww", "sh", "sw", "aux_dim", "has_logvar", "x3_exact")]
        + [("timestep_weight", C.c_float)]
        + [(n, C.c_int64) for n in ("kd", "kmlp", "kpe")]
        + [(n, C.c_void_p) for n in ("pe_w", "pe_b", "pos", "freqs", "aux_w", "aux_b", "l1_w", "l1_b", "l2_w", "l2_b",
                                     "mod_w", "mod_b", "logvar_w", "logvar_b", "head_w")]
        + [("layers_host", C.POINTER(Layer))]
    )


_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    "swiftk_version": ([], C.c_int),
    "swiftk_gemm_k_pad": ([_i, _l], _l),
    "swiftk_gemm": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _i, _i, _i, _p, _p, _l, _p], _i),
    "swiftk_gemm_chunked": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _i, _i, _i, _p, _p, _l, _i, _p, _l, _p], _i),
    "swiftk_gemm_chunk_scratch_bytes": ([], _l),
    "swiftk_window_attention": ([_p, _l, _p, _l, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_modnorm_residual": ([_p, _l, _p, _p, _l, _p, _p, _p, _l, _l, _i, _l, _f, _i, _p], _i),
    "swiftk_modnorm_residual_pair": ([_p, _l, _p, _l, _p, _l, _i, _p, _p, _p, _l, _l, _i, _l, _f, _p], _i),
    "swiftk_modnorm_residual_pair_to": ([_p, _l, _p, _p, _l, _p, _l, _i, _p, _p, _p, _l, _l, _i, _l, _f, _p], _i),
    "swiftk_modnorm_residual_pair_slabs": ([_p, _l, _l, _p, _l, _p, _l, _i, _p, _p, _p, _l, _l, _i, _l, _f, _p], _i),
    "swiftk_split_pair": ([_p, _l, _p, _l, _p, _l, _i, _l, _l, _p], _i),
    "swiftk_unit_noise": ([_p, _p, _p, _l, _i, _l, _i, _p], _i),
    "swiftk_counter_add": ([_p, _l, _p], _i),
    "swiftk_patchify": ([_p, _i, _f, _p, _i, _f, _p, _i, _f, _p, _l, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_unpatchify_affine": ([_p, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_timestep_embed": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p], _i),
    "swiftk_linear_small": ([_p, _l, _p, _l, _p, _p, _l, _i, _i, _i, _i, _p], _i),
    "swiftk_rollout_update": ([_p, _p, _p, _p, _p, _p, _i, _i, _l, _p], _i),
    "swiftk_cast_pad": ([_p, _l, _p, _l, _l, _l, _i, _p], _i),
    "swiftk_split3": ([_p, _l, _p, _l, _l, _l, _i, _p], _i),
    "swiftk_axpby": ([_p, _f, _p, _f, _p, _l, _p], _i),
    "swiftk_zero_f32": ([_p, _l, _p], _i),
    "swiftk_modnorm_residual_split3": ([_p, _l, _p, _p, _l, _p, _l, _p, _p, _p, _l, _l, _i, _l, _f, _p], _i),
    "swiftk_zero_check_report": ([_p], _i),
    "swiftk_unit_checksum": ([_p, _p, _p, _i, _l, _p], _i),
    "swiftk_timestep_embed_jvp": ([_p, _p, _p, _p, _i, _i, _f, _p], _i),
    "swiftk_silu_jvp": ([_p, _p, _p, _p, _l, _p], _i),
    "swiftk_qknorm_jvp": ([_p, _p, _l, _p, _p, _l, _i, _i, _i, _p], _i),
    "swiftk_window_attention_jvp": ([_p, _p, _l, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_modnorm_jvp": ([_p, _p, _l, _p, _p, _p, _p, _l, _p, _p, _p, _p, _l, _l, _i, _l, _f, _i, _p], _i),
    "swiftk_modnorm_jvp_pair": ([_p, _p, _l, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _l, _l, _i, _l, _f, _p], _i),
    "swiftk_swiglu_jvp": ([_p, _p, _l, _p, _p, _l, _l, _i, _i, _p], _i),
    "swiftk_gemm_bias_pos_pair": ([_p, _l, _p, _l, _p, _l, _p, _l, _l, _l, _l, _p, _p, _l, _p], _i),
    "swiftk_gemm_splitk_bf16": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _l, _i, _p], _i),
    "swiftk_gemm_tail_split_bf16": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _l, _p, _p], _i),
    "swiftk_modnorm_residual_pair_halves_bf16": ([_p, _l, _p, _p, _l, _p, _p, _p, _p, _l, _l, _i, _l, _f, _p], _i),
    "swiftk_modnorm_residual_pair_slabs_bf16": ([_p, _l, _l, _p, _l, _p, _l, _i, _p, _p, _p, _l, _l, _i, _l, _f, _p], _i),
    "swiftk_embed_bwd_sums": ([_p, _l, _p, _p, _p, _l, _l, _i, _l, _p], _i),
    "swiftk_cast_pad_t": ([_p, _l, _l, _l, _p, _l, _p, _l, _l, _p], _i),
    "swiftk_gemm_modnorm_residual_pair": ([_p, _l, _p, _l, _l, _p, _l, _p, _l, _p, _p, _p, _l, _l, _i, _l, _f, _i, _p], _i),
    "swiftk_gemm_jvp": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _i, _p, _p, _i, _p, _l, _p], _i),
    "swiftk_ensemble_sums": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_rmse_sums": ([_p, _p, _l, _p, _p, _i, _i, _i, _i, _p], _i),
    "swiftk_scm_target": ([_p, _p, _p, _p, _p, _f, _f, _p, _p, _i, _l, _p], _i),
    "swiftk_gemm_qkv_tiled": ([_p, _l, _p, _l, _p, _l, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_qkv_attention_fused": ([_p, _l, _p, _l, _p, _p, _l, _l, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_gemm_batched": ([_p, _l, _l, _p, _l, _l, _p, _l, _l, _i, _l, _l, _l, _i, _i, _p], _i),
    "swiftk_gemm_splitk": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _l, _i, _i, _p], _i),
    "swiftk_gemm_tn_splitk": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _l, _i, _p], _i),
    "swiftk_reduce_slabs": ([_p, _l, _l, _i, _p, _l, _l, _l, _i, _p], _i),
    "swiftk_transpose": ([_p, _l, _p, _l, _l, _l, _i, _p], _i),
    "swiftk_swiglu_fwd": ([_p, _l, _p, _l, _l, _i, _i, _p], _i),
    "swiftk_swiglu_bwd": ([_p, _l, _p, _l, _p, _l, _l, _i, _i, _p], _i),
    "swiftk_modnorm_bwd": ([_p, _l, _p, _p, _l, _p, _p, _p, _l, _p, _p, _p, _l, _p, _l, _i, _l, _f, _i, _p], _i),
    "swiftk_modnorm_bwd_ws0": ([_p, _l, _p, _p, _l, _p, _p, _p, _l, _p, _p, _p, _l, _p, _l, _i, _l, _f, _i, _p], _i),
    "swiftk_qknorm_bwd": ([_p, _p, _l, _p, _p, _l, _p, _p, _l, _i, _i, _i, _p], _i),
    "swiftk_window_attention_bwd_qknorm": ([_p, _l, _p, _p, _l, _p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_window_attention_bwd": ([_p, _l, _p, _p, _l, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_window_attention_bwd_scaled": ([_p, _l, _p, _p, _l, _p, _l, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "swiftk_colsum": ([_p, _l, _p, _l, _i, _l, _p], _i),
    "swiftk_linear_small_bwd": ([_p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _p, _i, _i, _i, _p], _i),
    "swiftk_silu_bwd": ([_p, _p, _p, _l, _p], _i),
    "swiftk_crps_loss": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p], _i),
    "swiftk_trigflow_prep": ([_p, _p, _p, _p, _p, _f, _i, _l, _p], _i),
    "swiftk_trigflow_loss": ([_p, _p, _p, _p, _p, _p, _p, _p, _f, _i, _i, _i, _i, _f, _p], _i),
    "swiftk_axpby_per_sample": ([_p, _p, _p, _p, _p, _i, _l, _p], _i),
    "swiftk_channel_axpy": ([_p, _p, _p, _p, _i, _i, _l, _p], _i),
    "swiftk_adamw_ema_step": ([_p, _i, _p, _p, _p, C.POINTER(OptHyper), _p], _i),
    "swiftk_profile_gemm": ([_i, _l], _i),
    "swiftk_set_tuning": ([_i, _i], _i),
    "swiftk_get_tuning": ([_i], _i),
    "swiftk_profile_collect": ([C.POINTER(C.c_double), C.POINTER(C.c_int64)], _i),
    "swiftk_workspace_bytes": ([C.POINTER(Model), _i], _l),
    "swiftk_swinv2_forward": ([C.POINTER(Model), _p, _i, _f, _p, _i, _f, _p, _i, _f, _p, _p, _p, _p, _p, _p, _p, _i, _p,
                               _l, _p], _i),
}
EXPORTS = tuple(_SIGS)

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwiftkError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C swift_amd/csrc)")
        h = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(h, name)
            fn.argtypes, fn.restype = args, res
        # A/B knobs for measurements (include/swiftk.h: swiftk_set_tuning), e.g. SWIFTK_TUNE=13:512,8:0 -- applied once, at load,
        # so that every entry point of the package (bench, tests, CLIs, tools) runs the same variant
        for kv in filter(None, os.environ.get("SWIFTK_TUNE", "").split(",")):
            k, v = (int(x) for x in kv.split(":"))
            if h.swiftk_set_tuning(k, v) != 0:
                raise SwiftkError(f"SWIFTK_TUNE: unknown tuning key {k}")
        _lib = h
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = _ERR.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise SwiftkError(f"{what} failed: {msg}")
