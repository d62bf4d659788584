// swiftk_swinv2_forward: the whole SwinV2 denoiser as one C call (launch sequence on one HIP stream).
// Host-side orchestration only; every arithmetic step is one of the kernels in gemm/attention/elementwise.hip.
// Mirrors reference src/swift/models/swinv2.py:305-330 (+ precond.py:139-148, diffusion.py:459 folded in).
#include "common.h"

int g_fwd_tiled = 1;  // tuning key 5 (A/B only): 0 keeps q/k/v row-major between to_qkv and attention
int g_x3_exact = 17;  // tuning key 11: the DEFAULT mask a host binding copies into swiftk_model.x3_exact when it packs the weights
                      // (the forward reads the model's own field, never this global: ADVICE r3)
int g_fwd_pair = 2;   // tuning key 12: bf16 engine's residual stream: 2 = (bf16 hi, 8-bit lo) pair, 1 = (bf16 hi, bf16 lo), 0 = fp32 + copy
int g_f32_chunk_k = 256;  // tuning key 13: fp32-operand GEMMs of the forward accumulate in chains of this many k (0 = one chain over K)
int g_fwd_splitk = 2;  // tuning key 14: 2 = bf16 slabs, 1 = fp32 slabs
int g_fwd_tail = 3;    // tuning key 29: bit 0 = wo / w2 with the walk's last round as two k-halves where that round is at most half full
                       // (3, 4, 6, 11, 12 .. units per step at dim 1056), bit 1 = the packed two-slab norm behind the one-unit split-K
int g_fwd_pepair = 1;  // tuning key 19: the patch embedding's epilogue writes the pair form itself
int g_x3_attnpv = 1;  // tuning key 28: split engine, P V of the fp32 attention kernel as three bf16 products
int g_x3_qkonly = 2;  // tuning key 27: split engine's exact to_qkv recompute: 2 = each hot head alone (q, k, v), 1 = hot pairs' q and k, 0 = hot pairs
int g_x3_ffsplit = 1;  // tuning key 18: split engine, w1 writes w2's operand blocks itself  // tuning key 14: wo / w2 as two k-ranges into fp32 slabs when their tiles fill less than half the chip (one unit per step)
int g_fwd_fused = 1;  // tuning key 8 (A/B only): 0 = to_qkv and window attention as two kernels (q/k/v window-tiled through HBM)

namespace {

__global__ __launch_bounds__(256) void zero_cols_kernel(char* p, int64_t ld_b, int64_t col0_b, int64_t ncol_b, int64_t rows) {
    // zero bytes [col0_b, col0_b + ncol_b) of every row (4-byte granularity)
    const int64_t per = ncol_b >> 2, total = rows * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per, c = i - r * per;
        *reinterpret_cast<uint32_t*>(p + r * ld_b + col0_b + 4 * c) = 0u;
    }
}

struct Layout {
    int64_t emb, h1, lat, mod, ape, x, xt, xlo, qkv, att, y, yslab, hmid, tok, kscr, a3, a3h, total;
};

inline int64_t al(int64_t v) { return (v + 255) & ~(int64_t)255; }

bool model_ok(const swiftk_model* m) {
    if (!m || !m->layers_host) return false;
    if (m->dtype != SWIFTK_F32 && m->dtype != SWIFTK_BF16 && m->dtype != SWIFTK_BF16X3) return false;
    if (m->depth <= 0 || m->dim <= 0 || m->heads <= 0 || m->dim % m->heads) return false;
    if (m->H % m->p1 || m->W % m->p2) return false;
    if (m->wh != 16 || m->ww != 16) return false;
    if ((m->H / m->p1) % 16 || (m->W / m->p2) % 16) return false;
    if (m->dim % 4 || m->mlp % 2) return false;
    return true;
}

// bf16 engine, pair-form stream: wo / w2 have (M / 256) x (d / 352) output tiles -- 96 per unit for 256 CUs, so one unit fills
// 37 % of a round of the persistent grid.  While two k-ranges per tile still fit ONE round the two GEMMs run as split-K into two
// slabs which the norm kernel sums on its way: B = 1 235 -> 250 sample-steps/s with fp32 slabs (255 replayed as a graph), 259 -> 277
// with bf16 slabs (key 14 = 2, the default).  Not beyond one round: at three units (288 tiles, 56 % of two rounds) the split fills
// 75 % of three, but it loses with either slab type (fp32: 306 -> 272; bf16: 324 -> 310, measured).
inline bool small_m_splitk(const swiftk_model* m, int64_t M) {
    const int64_t tiles = ((M + 255) / 256) * ((m->dim + 351) / 352);
    if (!(m->dtype == SWIFTK_BF16 && g_fwd_splitk && g_fwd_pair && m->dim % 8 == 0)) return false;
    if (2 * tiles <= 256) return true;
    // (experiment, key 14 = 3: with bf16 slabs also where two half-depth items per tile fill their rounds >= 1.25 x better --
    // measured at three / four units per step: 324 -> 310 / 356 -> 332 sample-steps/s: the half-depth k-loops pay two epilogues)
    if (g_fwd_splitk == 3 && tiles <= 1024) {
        const double plain = (double)tiles / (double)(((tiles + 255) / 256) * 256);
        const double split = (double)(2 * tiles) / (double)(((2 * tiles + 255) / 256) * 256);
        return split >= 1.25 * plain;
    }
    return false;
}

// wo / w2 of the bf16 pair engine with the persistent walk's last round as k-halves (swiftk_gemm_tail_split_bf16): the shapes it takes
// (352-wide tiles, 8-bit low parts, the packed norm's widths); the GEMM entry itself decides whether M leaves such a round
inline bool tail_split_shape(const swiftk_model* m, int64_t M) {
    if (!(m->dtype == SWIFTK_BF16 && (g_fwd_tail & 1) && g_fwd_pair >= 2 && m->dim == 1056 && M % 16 == 0)) return false;
    const int64_t tiles = ((M + 255) / 256) * (m->dim / 352), r = tiles % g_persist_wgs;
    return tiles > g_persist_wgs && r > 0 && 2 * r <= g_persist_wgs;
}

Layout make_layout(const swiftk_model* m, int B) {
    const int64_t es = m->dtype == SWIFTK_BF16 ? 2 : 4;
    const int64_t ntok = (int64_t)(m->H / m->p1) * (m->W / m->p2), M = ntok * B, d = m->dim;
    Layout L;
    int64_t o = 0;
    L.emb = o; o += al(B * d * 4);
    L.h1 = o; o += al(B * d * 4);
    L.lat = o; o += al(B * d * 4);
    L.mod = o; o += al((int64_t)B * m->depth * 4 * d * 4);
    L.ape = o; o += al(M * m->kpe * es);
    L.x = o; o += al(M * d * 4);
    L.xt = o; o += al(M * m->kd * es);
    L.xlo = o; o += m->dtype == SWIFTK_BF16 ? al(M * d * 2) : 0;  // low half of the pair-form residual stream
    L.qkv = o; o += al(M * 3 * d * es);
    L.att = o; o += al(M * m->kd * es);
    L.y = o; o += al(M * d * es);
    L.yslab = o; o += small_m_splitk(m, M) ? al(2 * M * d * 4 + 8192) : tail_split_shape(m, M) ? al(2 * M * d * 2 + 8192) : 0;  // two fp32 slabs of the split-K wo / w2 (one unit per step)
    L.hmid = o; o += al(M * m->kmlp * es);
    L.tok = o; o += al(M * (int64_t)((m->out_ch * m->p1 * m->p2 + 3) & ~3) * 4);
    L.kscr = o;
    // parked accumulators of the two-level fp32 GEMMs: the exact engine only (the split engine's exact-kernel GEMMs run one
    // chain; sized whatever tuning key 13 says now, since the key may change after the workspace was sized)
    if (m->dtype == SWIFTK_F32) o += al(swiftk_gemm_chunk_scratch_bytes());
    L.a3 = o;
    if (m->dtype == SWIFTK_BF16X3) {  // the split GEMM operand [M, k_pad(3 K)] bf16 of the widest GEMM input
        const int64_t kin = (int64_t)m->in_ch * m->p1 * m->p2;
        const int64_t kmax = kin > m->mlp ? (kin > d ? kin : d) : (m->mlp > d ? m->mlp : d);
        o += al(M * swiftk_gemm_k_pad(SWIFTK_BF16, 3 * ((kmax + 3) & ~(int64_t)3)) * 2);
    }
    L.a3h = o;  // ... and the hidden activation's blocks, written by w1's epilogue while w1 still reads its own operand from a3
    if (m->dtype == SWIFTK_BF16X3) o += al(M * swiftk_gemm_k_pad(SWIFTK_BF16, 3 * (((int64_t)m->mlp + 3) & ~(int64_t)3)) * 2);
    L.total = o;
    return L;
}

}  // namespace

#define RUN(expr)                  \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != 0) return rc__; \
    } while (0)

extern "C" int64_t swiftk_workspace_bytes(const swiftk_model* m, int B) {
    if (!model_ok(m) || B <= 0) return 0;
    return make_layout(m, B).total;
}

extern "C" int swiftk_swinv2_forward(const swiftk_model* m, const float* src0, int c0, float s0, const float* src1, int c1,
                                     float s1, const float* src2, int c2, float s2, const float* t, const float* aux,
                                     const float* xt, const float* alpha, const float* beta, float* out, float* logvar,
                                     int B, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!model_ok(m)) return SWIFTK_ESHAPE;
    if (!src0 || !t || !out || !workspace || B <= 0) return SWIFTK_EINVAL;
    if (c0 + c1 + c2 != m->in_ch) return SWIFTK_ESHAPE;
    if (B > SWIFTK_MAX_UNITS) return SWIFTK_ESHAPE;  // M = B * tokens rows must stay below 2^30 (int tile coordinates)
    if ((uintptr_t)workspace & 255) return SWIFTK_EALIGN;
    const Layout L = make_layout(m, B);
    if (workspace_bytes < L.total) return SWIFTK_EWORKSPACE;
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // SWIFTK_BF16X3: every kernel but the GEMMs runs exactly as in the fp32 engine; a GEMM's fp32 input is first split into
    // [hi | lo | hi] bf16 blocks (swiftk_split3) and multiplied with the [hi | hi | lo] weight on the bf16 MFMA kernel
    const bool x3 = m->dtype == SWIFTK_BF16X3;
    const int x3_exact = m->x3_exact;  // which GEMMs keep fp32 operands: fixed when the weights were packed
    const int dt = x3 ? SWIFTK_F32 : m->dtype;
    const int64_t es = dt == SWIFTK_BF16 ? 2 : 4;
    const int gh = m->H / m->p1, gw = m->W / m->p2, d = m->dim, hd = m->dim / m->heads;
    const int64_t ntok = (int64_t)gh * gw, M = ntok * B;
    const int64_t ldmod = (int64_t)m->depth * 4 * d;
    float* emb = reinterpret_cast<float*>(ws + L.emb);
    float* h1 = reinterpret_cast<float*>(ws + L.h1);
    float* lat = reinterpret_cast<float*>(ws + L.lat);
    float* mod = reinterpret_cast<float*>(ws + L.mod);
    float* x = reinterpret_cast<float*>(ws + L.x);
    float* tok = reinterpret_cast<float*>(ws + L.tok);
    void *ape = ws + L.ape, *xT = ws + L.xt, *qkv = ws + L.qkv, *att = ws + L.att, *y = ws + L.y, *hmid = ws + L.hmid;
    void* a3 = ws + L.a3;
    // split engine, FeedForward: w1's epilogue writes the hidden activation straight as w2's operand blocks (no fp32 h, no split pass)
    const int64_t kvh = ((int64_t)m->mlp + 3) & ~(int64_t)3, ld3h = swiftk_gemm_k_pad(SWIFTK_BF16, 3 * kvh);
    const bool ff_split = x3 && !(x3_exact & 12) && kvh == m->mlp && m->mlp % 8 == 0 && g_x3_ffsplit;
    char* a3h = ws + L.a3h;
    // C = epilogue(A W^T) over the M token rows: `kpass` is the K handed to swiftk_gemm in the one-product engines (the padded
    // width, or the valid one when it ends half-way into the last k-tile), `kvalid` the number of meaningful columns of A
    const void* a3_holds = nullptr;  // split engine: the fp32 tensor whose [hi | lo | hi] blocks currently sit in a3 (d columns), if any
    // x += ModulatedNorm(y) of the fp32 / split engines.  Split engine: the norm writes the next GEMM's operand blocks into a3 in the
    // same pass (tuning key 26) -- and no fp32 operand copy when `need_copy` is false (nobody but that GEMM reads it)
    auto NORM = [&](const void* yv, const float* g_, const float* b_, const float* mod_, bool need_copy) -> int {
        if (x3 && g_x3_normsplit && (d & 3) == 0) {
            const int64_t ld3 = swiftk_gemm_k_pad(SWIFTK_BF16, 3 * (int64_t)d);
            const int rc = swiftk_modnorm_residual_split3(static_cast<const float*>(yv), d, x, need_copy ? static_cast<float*>(xT) : nullptr, m->kd, a3,
                                                          ld3, g_, b_, mod_, ldmod, M, d, ntok, 1e-6f, stream);
            if (rc == 0) {
                a3_holds = xT;
                return 0;
            }
            if (rc != SWIFTK_ESHAPE) return rc;
        }
        a3_holds = nullptr;
        return swiftk_modnorm_residual(yv, d, x, xT, m->kd, g_, b_, mod_, ldmod, M, d, ntok, 1e-6f, dt, stream);
    };
    auto G = [&](const void* A, int64_t lda, const void* Wm, void* Cm, int64_t ldc, int64_t N, int64_t kpass, int64_t kvalid,
                 int out_dt, int epi, const float* e0, const float* e1, int64_t pr, bool exact = false) -> int {
        // the exact-fp32 engine's products: two-level accumulation (gemm.hip).  The split engine's two exact GEMMs keep one chain:
        // its error is the (hi, lo) split's 2^-17, not the chain's
        if (dt == SWIFTK_F32 && !x3 && g_f32_chunk_k > 0)
            return swiftk_gemm_chunked(A, lda, Wm, lda, Cm, ldc, M, N, kpass, dt, out_dt, epi, e0, e1, pr, g_f32_chunk_k, ws + L.kscr,
                                       swiftk_gemm_chunk_scratch_bytes(), stream);
        if (!x3 || exact) return swiftk_gemm(A, lda, Wm, lda, Cm, ldc, M, N, kpass, dt, out_dt, epi, e0, e1, pr, stream);
        const int64_t kv = (kvalid + 3) & ~(int64_t)3;  // (columns [kvalid, kv) of A are zero k-padding; the weight has them too)
        const int64_t ld3 = swiftk_gemm_k_pad(SWIFTK_BF16, 3 * kv);
        // (a3_holds: the ModulatedNorm that produced A wrote its operand blocks into a3 itself -- swiftk_modnorm_residual_split3)
        if (!(a3_holds == A && kv == d)) RUN(swiftk_split3(static_cast<const float*>(A), lda, a3, ld3, M, kv, 0, stream));
        a3_holds = nullptr;  // (this GEMM's caller may overwrite a3 next)
        const int64_t k3 = ((3 * kv) % 64 == 32 && ld3 >= 3 * kv + 32) ? 3 * kv : ld3;
        return swiftk_gemm(a3, ld3, Wm, ld3, Cm, ldc, M, N, k3, SWIFTK_BF16, SWIFTK_F32, epi, e0, e1, pr, stream);
    };

    // time / auxiliary embedding -> latent -> all 2*depth modulation vectors in one pass (swinv2.py:316-321, :85)
    RUN(swiftk_timestep_embed(t, m->aux_dim > 0 ? aux : nullptr, m->freqs, m->aux_w, m->aux_b, emb, B, d, m->aux_dim,
                              m->timestep_weight, stream));
    RUN(swiftk_linear_small(emb, d, m->l1_w, d, m->l1_b, h1, d, B, d, d, 1, stream));
    RUN(swiftk_linear_small(h1, d, m->l2_w, d, m->l2_b, lat, d, B, d, d, 1, stream));
    // all 2 x depth modulation Linears as one [4 depth d, d] fp32 matrix (214 MB at Swift-B).  From 16 samples on it runs on the
    // fp32 MFMA GEMM (exact fp32 FMA chains, bias in the epilogue): the VALU kernel walks the matrix once per 8 samples
    // (12 passes, 1.4 ms at 96 units), the GEMM streams it once (one 256-row tile row, 144 column tiles)
    if (B >= 16 && d % 32 == 0 && ldmod % 4 == 0 &&
        swiftk_gemm(lat, d, m->mod_w, d, mod, ldmod, B, ldmod, d, SWIFTK_F32, SWIFTK_F32, SWIFTK_EPI_BIAS_POS, m->mod_b, nullptr, 0,
                    stream) == 0) {
    } else {
        RUN(swiftk_linear_small(lat, d, m->mod_w, d, m->mod_b, mod, ldmod, B, (int)ldmod, d, 0, stream));
    }
    if (logvar) {
        if (!m->logvar_w) return SWIFTK_EINVAL;
        RUN(swiftk_linear_small(lat, d, m->logvar_w, d, m->logvar_b, logvar, 1, B, 1, d, 0, stream));
    }

    // patch embedding (+bias +pos_embed) into the fp32 residual stream, plus its GEMM-operand copy
    RUN(swiftk_patchify(src0, c0, s0, src1, c1, s1, src2, c2, s2, ape, m->kpe, B, m->H, m->W, m->p1, m->p2, dt, stream));
    // bf16 engine: the residual stream is the pair (xT = hi, xlo = lo) from the patch embedding on; the fp32 x is never formed when
    // the embedding's epilogue can write the pair itself (8-bit low parts, whole 16-column groups), else x is split once here
    const bool pair = dt == SWIFTK_BF16 && g_fwd_pair && ntok % 16 == 0 && d % 8 == 0 && d <= 2048;
    void* xlo = ws + L.xlo;
    const int lo_bits = g_fwd_pair == 1 ? 16 : 8;
    const bool splitk = pair && small_m_splitk(m, M);
    const bool tail = pair && lo_bits == 8 && !splitk && tail_split_shape(m, M);
    // stride between the two bf16 slabs: M d elements would put slab 1 a multiple of 16.5 MiB behind slab 0 -- the norm kernel's two
    // requests per chunk would meet the same memory channels; 17 x 256 B more spreads them
    const int64_t SS = M * d + 2176;
    const bool halves_norm = (g_fwd_tail & 2) && lo_bits == 8 && (d == 1056 || d == 1280) && M % 16 == 0;  // packed two-slab norm
    float* yslab = reinterpret_cast<float*>(ws + L.yslab);
    const bool pe_pair = pair && lo_bits == 8 && d % 16 == 0 && m->kpe % 64 == 0 && g_fwd_pepair;
    if (pe_pair) {
        if (m->kd > d) {  // (hi is a GEMM operand: its k-padding columns are zeroed once per evaluation)
            hipLaunchKernelGGL(zero_cols_kernel, dim3(1024), dim3(256), 0, st, static_cast<char*>(xT), m->kd * 2, (int64_t)d * 2,
                               (m->kd - d) * 2, M);
            SWIFTK_CHECK_LAUNCH();
        }
        RUN(swiftk_gemm_bias_pos_pair(ape, m->kpe, m->pe_w, m->kpe, xT, m->kd, xlo, d, M, d, m->kpe, m->pe_b, m->pos, ntok, stream));
    } else {
        RUN(G(ape, m->kpe, m->pe_w, x, d, d, m->kpe, (int64_t)m->in_ch * m->p1 * m->p2, SWIFTK_F32, SWIFTK_EPI_BIAS_POS, m->pe_b, m->pos,
              ntok, (x3_exact & 16) != 0));
        if (pair) RUN(swiftk_split_pair(x, d, xT, m->kd, xlo, d, lo_bits, M, d, stream));
        else RUN(swiftk_cast_pad(x, d, xT, m->kd, M, d, dt, stream));
    }
    if (m->kd > d) {  // K-padding columns of the attention output must be finite (they meet zero weight columns)
        hipLaunchKernelGGL(zero_cols_kernel, dim3(1024), dim3(256), 0, st, static_cast<char*>(att), m->kd * es, d * es,
                           (m->kd - d) * es, M);
        SWIFTK_CHECK_LAUNCH();
    }
    if (ff_split && ld3h > 3 * kvh) {  // k-padding behind the three blocks (the epilogue writes the blocks only)
        hipLaunchKernelGGL(zero_cols_kernel, dim3(1024), dim3(256), 0, st, a3h, ld3h * 2, 3 * kvh * 2, (ld3h - 3 * kvh) * 2, M);
        SWIFTK_CHECK_LAUNCH();
    }
    if (m->kmlp > m->mlp) {
        hipLaunchKernelGGL(zero_cols_kernel, dim3(1024), dim3(256), 0, st, static_cast<char*>(hmid), m->kmlp * es,
                           (int64_t)m->mlp * es, (m->kmlp - m->mlp) * es, M);
        SWIFTK_CHECK_LAUNCH();
    }

    // valid K of the d-wide GEMMs: when it ends half-way into the last 128-B k-tile (1056 bf16 = 16.5 tiles) the GEMM
    // skips the padded half; otherwise it is the padded width itself
    const int64_t half_tile = (dt == SWIFTK_BF16 ? 64 : 32) / 2;
    const int64_t kdv = (m->kd - d == half_tile) ? d : m->kd;
    const bool do_shift = (m->sh != 0) || (m->sw != 0);
    // small batches, bf16 engine: wo / w2 and their ModulatedNorm as one complete-row kernel (gemm_rownorm.hip) -- 32-row
    // workgroups while those fill at most one round of the persistent grid's CUs, 64-row ones beyond
    int rn_rows = 0;
    if (pair && lo_bits == 8 && B <= g_fwd_rownorm && (d == 1056 || d == 960) && kdv % 32 == 0 && m->kmlp % 32 == 0) {
        rn_rows = (M / 32 <= g_persist_wgs) ? 32 : 64;
        if (ntok % rn_rows) rn_rows = 0;
    }
    int64_t tail3[3] = {0, 0, 8};  // swiftk_gemm_tail_split_bf16's description of its walk, for the norm behind it
    int tail_rc = 0;
    for (int i = 0; i < m->depth; ++i) {
        const swiftk_layer& ly = m->layers_host[i];
        const bool shifted = do_shift && (i & 1);
        // cosine-attention's normalise/scale prologue rides in the to_qkv epilogue (fp32 accumulators): head_dim 88, and
        // 80 / 96 (the 468 M / 664 M variants) with an even head count, either operand type
        const bool fuse_norm = (hd == 88) || ((hd == 80 || hd == 96) && m->heads % 2 == 0 && M % 8 == 0);
        // bf16 operands, head_dim 80 / 88 / 96 (the 468 M variant, Swift-B, the 664 M variant): to_qkv, the cosine norm and the
        // window attention run as ONE kernel (q/k/v stay on the CU)
        if (dt == SWIFTK_BF16 && (hd == 80 || hd == 88 || hd == 96) && g_fwd_fused && m->wh == 16 && m->ww == 16) {
            RUN(swiftk_qkv_attention_fused(xT, m->kd, ly.qkv_w, m->kd, ly.scale, att, m->kd, kdv, B, gh, gw, m->heads, hd,
                                           shifted ? m->sh : 0, shifted ? m->sw : 0, stream));
        } else if (fuse_norm && dt == SWIFTK_BF16 && g_fwd_tiled) {
            // bf16: q/k/v leave the GEMM window-tiled, so every attention operand is one contiguous 44-KiB block
            RUN(swiftk_gemm_qkv_tiled(xT, m->kd, ly.qkv_w, m->kd, qkv, kdv, ly.scale, B, gh, gw, m->heads, hd,
                                      shifted ? m->sh : 0, shifted ? m->sw : 0, stream));
            RUN(swiftk_window_attention(qkv, 3 * d, att, m->kd, ly.scale, B, gh, gw, m->heads, hd, shifted ? m->sh : 0,
                                        shifted ? m->sw : 0, dt, SWIFTK_ATTN_PRENORM | SWIFTK_ATTN_TILED, stream));
        } else {
            // (SWIFTK_BF16X3: the cosine logits multiply q-hat . k-hat by up to 100, so the split product's 4.5e-6 can reach the
            // softmax as 4.5e-4 -- x3_exact bit 0 keeps the whole GEMM on the exact-fp32 kernel, bit 6 only the hot head pairs)
            RUN(G(xT, m->kd, ly.qkv_w, qkv, 3 * d, 3 * d, kdv, d, dt, fuse_norm ? SWIFTK_EPI_QKNORM : SWIFTK_EPI_NONE,
                  fuse_norm ? ly.scale : nullptr, nullptr, fuse_norm ? hd : 0, (x3_exact & 1) != 0));
            // (bit 6 without bit 0 promises the hot-pair recompute: a model it cannot serve -- no fused norm, or hot pairs
            // without their fp32 weights -- is refused rather than run fully split)
            if (x3 && (x3_exact & 64) && !(x3_exact & 1) && (!fuse_norm || (ly.qk_exact_pairs && !ly.qkv_w_f32)))
                return SWIFTK_EINVAL;
            if (x3 && (x3_exact & 64) && !(x3_exact & 1) && ly.qkv_w_f32) {
                // adaptive to_qkv of the split engine: the head pairs whose logit scale is large enough for the split product's
                // 4.5e-6 to matter in front of the softmax are recomputed on the exact-fp32 kernel -- 6 head_dim output columns each,
                // written over the split result (same QK-norm epilogue, the pair's two logit scales)
                // (round 6) the packer may name the hot HEADS themselves (bits 16..31 of qk_exact_pairs, models of up to 16 heads): then each
                // is recomputed alone -- its [q | k | v], N = 3 head_dim, is ONE 352-wide tile column where the pair's [q | k | v] x 2 takes
                // two (427 against 840 us at 8 units) -- and the pair's cold head keeps the split product like every other cold head.
                // Tuning key 27: 2 = per hot head (default), 1 = the pair's q and k columns only (the same cost, but the hot head's v is
                // then a split product: 9.9e-5 instead of 8.9e-5 on the Swift-B golden -- the v of a peaked softmax is not averaged
                // down over 256 keys), 0 = whole pairs
                const int hot_heads = (ly.qk_exact_pairs >> 16) & 0xffff;
                if (g_x3_qkonly >= 2 && hot_heads && m->heads <= 16) {
                    for (int h = 0; h < m->heads; ++h) {
                        if (!((hot_heads >> h) & 1)) continue;
                        const int64_t c0 = (int64_t)h * 3 * hd;
                        RUN(swiftk_gemm(xT, m->kd, static_cast<const float*>(ly.qkv_w_f32) + c0 * m->kd, m->kd, static_cast<float*>(qkv) + c0, 3 * d, M,
                                        3 * hd, kdv, SWIFTK_F32, SWIFTK_F32, SWIFTK_EPI_QKNORM, ly.scale + h, nullptr, hd, stream));
                    }
                } else
                for (int pp = 0; 2 * pp < m->heads; ++pp) {
                    if (!((ly.qk_exact_pairs >> pp) & 1)) continue;
                    const int64_t c0 = (int64_t)pp * 6 * hd;
                    // key 27 = 1: only q-hat and k-hat meet in the logits: the pair's two [q | k] column pairs (4 head_dim columns = ONE
                    // 352-wide tile column) are recomputed; v keeps its split product (pos_rows < 0 = the [q | k]-only form of QKNORM)
                    if (g_x3_qkonly == 1 &&
                        swiftk_gemm(xT, m->kd, static_cast<const float*>(ly.qkv_w_f32) + c0 * m->kd, m->kd, static_cast<float*>(qkv) + c0, 3 * d, M,
                                    4 * hd, kdv, SWIFTK_F32, SWIFTK_F32, SWIFTK_EPI_QKNORM, ly.scale + 2 * pp, nullptr, -(int64_t)hd, stream) == 0)
                        continue;
                    RUN(swiftk_gemm(xT, m->kd, static_cast<const float*>(ly.qkv_w_f32) + c0 * m->kd, m->kd,
                                    static_cast<float*>(qkv) + c0, 3 * d, M, 6 * hd, kdv, SWIFTK_F32, SWIFTK_F32, SWIFTK_EPI_QKNORM,
                                    ly.scale + 2 * pp, nullptr, hd, stream));
                }
            }
            // (round 6 also let the fp32 attention kernel write wo's [hi | lo | hi] operand blocks itself -- 8 B per element and a launch less:
            // 100.0 against 99.9 sample-steps/s, nothing; taken out again)
            RUN(swiftk_window_attention(qkv, 3 * d, att, m->kd, ly.scale, B, gh, gw, m->heads, hd, shifted ? m->sh : 0,
                                        shifted ? m->sw : 0, dt, (fuse_norm ? SWIFTK_ATTN_PRENORM : 0) | (x3 && g_x3_attnpv ? SWIFTK_ATTN_PV_BF16X3 : 0),
                                        stream));
        }
        if (rn_rows) {
            RUN(swiftk_gemm_modnorm_residual_pair(att, m->kd, ly.wo_w, m->kd, kdv, xT, m->kd, xlo, d, ly.ln1_g, ly.ln1_b,
                                                  mod + (int64_t)(2 * i) * 2 * d, ldmod, M, d, ntok, 1e-6f, rn_rows, stream));
        } else if (tail && (tail_rc = swiftk_gemm_tail_split_bf16(att, m->kd, ly.wo_w, m->kd, yslab, d, SS, M, d, kdv, tail3, stream)) != SWIFTK_ESHAPE) {
            // (SWIFTK_ESHAPE: the GEMM takes no such walk right now -- ping-pong loop or persistent kernel switched off by a tuning key -- and has
            // launched nothing: the plain path below runs instead)
            RUN(tail_rc);
            RUN(swiftk_modnorm_residual_pair_halves_bf16(yslab, SS, tail3, xT, m->kd, xlo, ly.ln1_g, ly.ln1_b,
                                                         mod + (int64_t)(2 * i) * 2 * d, ldmod, M, d, ntok, 1e-6f, stream));
        } else if (splitk && g_fwd_splitk >= 2) {
            RUN(swiftk_gemm_splitk_bf16(att, m->kd, ly.wo_w, m->kd, yslab, d, SS, M, d, kdv, 2, stream));
            if (halves_norm)
                RUN(swiftk_modnorm_residual_pair_halves_bf16(yslab, SS, nullptr, xT, m->kd, xlo, ly.ln1_g, ly.ln1_b,
                                                             mod + (int64_t)(2 * i) * 2 * d, ldmod, M, d, ntok, 1e-6f, stream));
            else
            RUN(swiftk_modnorm_residual_pair_slabs_bf16(yslab, d, SS, xT, m->kd, xlo, d, lo_bits, ly.ln1_g, ly.ln1_b,
                                                        mod + (int64_t)(2 * i) * 2 * d, ldmod, M, d, ntok, 1e-6f, stream));
        } else if (splitk) {
            RUN(swiftk_gemm_splitk(att, m->kd, ly.wo_w, m->kd, yslab, d, M * d, M, d, kdv, SWIFTK_BF16, 2, stream));
            RUN(swiftk_modnorm_residual_pair_slabs(yslab, d, M * d, xT, m->kd, xlo, d, lo_bits, ly.ln1_g, ly.ln1_b,
                                                   mod + (int64_t)(2 * i) * 2 * d, ldmod, M, d, ntok, 1e-6f, stream));
        } else {
            RUN(G(att, m->kd, ly.wo_w, y, d, d, kdv, d, dt, SWIFTK_EPI_NONE, nullptr, nullptr, 0, (x3_exact & 2) != 0));
            if (pair)
                RUN(swiftk_modnorm_residual_pair(y, d, xT, m->kd, xlo, d, lo_bits, ly.ln1_g, ly.ln1_b, mod + (int64_t)(2 * i) * 2 * d,
                                                 ldmod, M, d, ntok, 1e-6f, stream));
            else  // (split engine with the fused FeedForward: w1 is the only reader of this norm's output -- no fp32 operand copy)
                RUN(NORM(y, ly.ln1_g, ly.ln1_b, mod + (int64_t)(2 * i) * 2 * d, !ff_split));
        }
        if (ff_split) {
            const int64_t kv = (d + 3) & ~(int64_t)3, ld3 = swiftk_gemm_k_pad(SWIFTK_BF16, 3 * kv);
            if (a3_holds != xT) RUN(swiftk_split3(static_cast<const float*>(xT), m->kd, a3, ld3, M, kv, 0, stream));
            a3_holds = nullptr;
            const int64_t k3 = ((3 * kv) % 64 == 32 && ld3 >= 3 * kv + 32) ? 3 * kv : ld3;
            RUN(swiftk_gemm(a3, ld3, ly.w1_w, ld3, a3h, ld3h, M, 2 * m->mlp, k3, SWIFTK_BF16, SWIFTK_BF16, SWIFTK_EPI_SWIGLU_SPLIT3, nullptr,
                            nullptr, kvh, stream));
            const int64_t k3h = ((3 * kvh) % 64 == 32 && ld3h >= 3 * kvh + 32) ? 3 * kvh : ld3h;
            RUN(swiftk_gemm(a3h, ld3h, ly.w2_w, ld3h, y, d, M, d, k3h, SWIFTK_BF16, SWIFTK_F32, SWIFTK_EPI_NONE, nullptr, nullptr, 0, stream));
            RUN(NORM(y, ly.ln2_g, ly.ln2_b, mod + (int64_t)(2 * i + 1) * 2 * d, true));  // (the next to_qkv's hot pairs / the head read xT)
            continue;
        }
        RUN(G(xT, m->kd, ly.w1_w, hmid, m->kmlp, 2 * m->mlp, kdv, d, dt, SWIFTK_EPI_SWIGLU, nullptr, nullptr, 0, (x3_exact & 4) != 0));
        if (rn_rows) {
            RUN(swiftk_gemm_modnorm_residual_pair(hmid, m->kmlp, ly.w2_w, m->kmlp, m->kmlp, xT, m->kd, xlo, d, ly.ln2_g, ly.ln2_b,
                                                  mod + (int64_t)(2 * i + 1) * 2 * d, ldmod, M, d, ntok, 1e-6f, rn_rows, stream));
            continue;
        }
        if (tail && (tail_rc = swiftk_gemm_tail_split_bf16(hmid, m->kmlp, ly.w2_w, m->kmlp, yslab, d, SS, M, d, m->kmlp, tail3, stream)) != SWIFTK_ESHAPE) {
            RUN(tail_rc);
            RUN(swiftk_modnorm_residual_pair_halves_bf16(yslab, SS, tail3, xT, m->kd, xlo, ly.ln2_g, ly.ln2_b,
                                                         mod + (int64_t)(2 * i + 1) * 2 * d, ldmod, M, d, ntok, 1e-6f, stream));
            continue;
        }
        if (splitk && g_fwd_splitk >= 2) {
            RUN(swiftk_gemm_splitk_bf16(hmid, m->kmlp, ly.w2_w, m->kmlp, yslab, d, SS, M, d, m->kmlp, 2, stream));
            if (halves_norm)
                RUN(swiftk_modnorm_residual_pair_halves_bf16(yslab, SS, nullptr, xT, m->kd, xlo, ly.ln2_g, ly.ln2_b,
                                                             mod + (int64_t)(2 * i + 1) * 2 * d, ldmod, M, d, ntok, 1e-6f, stream));
            else
            RUN(swiftk_modnorm_residual_pair_slabs_bf16(yslab, d, SS, xT, m->kd, xlo, d, lo_bits, ly.ln2_g, ly.ln2_b,
                                                        mod + (int64_t)(2 * i + 1) * 2 * d, ldmod, M, d, ntok, 1e-6f, stream));
            continue;
        }
        if (splitk) {
            RUN(swiftk_gemm_splitk(hmid, m->kmlp, ly.w2_w, m->kmlp, yslab, d, M * d, M, d, m->kmlp, SWIFTK_BF16, 2, stream));
            RUN(swiftk_modnorm_residual_pair_slabs(yslab, d, M * d, xT, m->kd, xlo, d, lo_bits, ly.ln2_g, ly.ln2_b,
                                                   mod + (int64_t)(2 * i + 1) * 2 * d, ldmod, M, d, ntok, 1e-6f, stream));
            continue;
        }
        RUN(G(hmid, m->kmlp, ly.w2_w, y, d, d, m->kmlp, m->mlp, dt, SWIFTK_EPI_NONE, nullptr, nullptr, 0, (x3_exact & 8) != 0));
        if (pair)
            RUN(swiftk_modnorm_residual_pair(y, d, xT, m->kd, xlo, d, lo_bits, ly.ln2_g, ly.ln2_b, mod + (int64_t)(2 * i + 1) * 2 * d, ldmod,
                                             M, d, ntok, 1e-6f, stream));
        else
            RUN(NORM(y, ly.ln2_g, ly.ln2_b, mod + (int64_t)(2 * i + 1) * 2 * d, true));
    }

    // the head's output width rounded up to the GEMM's N granularity (head_w carries zero rows there: 69 -> 72 for 1x1 patches)
    const int po4 = (m->out_ch * m->p1 * m->p2 + 3) & ~3;
    RUN(G(xT, m->kd, m->head_w, tok, po4, po4, kdv, d, SWIFTK_F32, SWIFTK_EPI_NONE, nullptr, nullptr, 0, (x3_exact & 32) != 0));
    RUN(swiftk_unpatchify_affine(tok, po4, xt, alpha, beta, out, B, m->out_ch, m->H, m->W, m->p1, m->p2, stream));
    return 0;
}
