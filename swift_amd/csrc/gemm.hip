// C[M,N] = epilogue(A[M,K] * W[N,K]^T) for gfx950: MFMA, LDS-DMA staged, 256x352 tiles.
//
// Why 256 x 352: every N on the Swift-B path is a multiple of 352 = 11*32
// (3168 = 9*352, 1056 = 3*352, 5632 = 16*352, 2816 = 8*352) and M = B*8192 is
// a multiple of 256, so no tile is wasted; 8 waves sit 4(M) x 2(N), each owning
// 64 x 176 = 4 x 11 MFMA 16x16 tiles (176 accumulator VGPRs).
//
// Data movement: one k-tile is 128 B per row for both operands (64 bf16 or 32
// fp32), brought HBM/L2 -> LDS by `global_load_lds_dwordx4` (no VGPR round
// trip), double buffered, one barrier per k-tile.  The LDS image is lane-linear
// per 1-KiB piece (8 rows x 128 B), so the bank-conflict swizzle
// (16-B chunk ^= (row>>1)&7) is applied to the per-lane SOURCE address and
// again on the fragment read (cdna_hip_programming.md section 5.4 rule 21).
//
// Operand roles are swapped (MFMA A := W rows, B := activation rows) so each
// lane ends up with 4 consecutive output columns of one row: 8-16 B stores and
// the SwiGLU (gate, up) pair in one lane.
//
// The same kernel serves fp32 (v_mfma_f32_16x16x4_f32, exact fp32 FMA chain)
// and bf16 (v_mfma_f32_16x16x32_bf16): a 16-B chunk is 4 fp32 (4 MFMAs) or
// 8 bf16 (1 MFMA) and everything else is byte-identical.
#include "common.h"

int g_persist_wgs = 256;  // tuning key 2: workgroups of the persistent matrix kernels (one per CU the stream may use)

namespace {

// silu(gate) * up in a GEMM epilogue.  bf16 output (the throughput engine): v_exp_f32 + v_rcp_f32, 1 ulp each, far below the
// output rounding.  fp32 output (the exact engine, judged at 1e-4 against the reference over 39 chained evaluations): libm expf
// and a true division, as ATen's silu -- the fp32 engine's GEMMs run on the 157 TF pipe, the extra VALU work hides there.
template <typename OutT>
__device__ __forceinline__ float swiglu_out(float g, float u) {
    if constexpr (sizeof(OutT) == 4) return g / (1.0f + expf(-g)) * u;
    else return g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)) * u;
}

constexpr int BM = 256;
constexpr int BN = 352;
constexpr int ROWB = 128;                 // bytes per tile row per k-tile
constexpr int A_BYTES = BM * ROWB;        // 32 KiB
constexpr int B_BYTES = BN * ROWB;        // 44 KiB
constexpr int STAGE = A_BYTES + B_BYTES;  // 76 KiB
constexpr int LDS_BYTES = 2 * STAGE;      // 152 KiB of the CU's 160 KiB
constexpr int NT = 512;
constexpr int A_PIECES = A_BYTES / 1024;  // 32 (4 per wave)
constexpr int B_PIECES = B_BYTES / 1024;  // 44 (5 or 6 per wave)
constexpr int MI = 4, NI = 11;  // NI: W-side MFMA tiles per wave of the 256 x 352 geometry (gemm_kernel); gemm_kernel_p takes it
                                // as a template parameter: 10 / 11 / 12 -> 320 / 352 / 384 columns per tile

constexpr int EPI_QKNORM_TILED = 4;  // internal: SWIFTK_EPI_QKNORM with the window-tiled store (swiftk_gemm_qkv_tiled)
constexpr int EPI_NONE_TAIL = 12;     // internal: SWIFTK_EPI_NONE with the LAST ROUND's tiles as two k-halves (swiftk_gemm_tail_split_bf16)
constexpr int EPI_BIAS_POS_PAIR = 11;  // internal: SWIFTK_EPI_BIAS_POS leaving as the (bf16 hi, 8-bit lo) pair (swiftk_gemm_bias_pos_pair)

// Build-time switches.  SWIFTK_GEMM_INSTR = 1 compiles the timing experiments (tuning key 3: ablation bits, s_memtime
// timeline) into the persistent kernel -- `make variant EXTRA=-DSWIFTK_GEMM_INSTR=1`, never into libswiftk.so.
#ifndef SWIFTK_GEMM_INSTR
#define SWIFTK_GEMM_INSTR 0
#endif
#ifndef SWIFTK_X_VMCNT
#define SWIFTK_X_VMCNT 1
#endif
#ifndef SWIFTK_X_PRIO
#define SWIFTK_X_PRIO 1
#endif
#ifndef SWIFTK_X_PF2
#define SWIFTK_X_PF2 1
#endif
// timing probe with WRONG results: read only every n-th W fragment from LDS (n = 2: 10 instead of 15 ds_read_b128 per 44 MFMAs,
// the ratio a one-wave-per-SIMD 128 x 176 register tile would have) -- does the plateau move with LDS reads per MFMA?
#ifndef SWIFTK_X_FEWREADS
#define SWIFTK_X_FEWREADS 0
#endif
#ifndef SWIFTK_X_NOSILU
#define SWIFTK_X_NOSILU 0
#endif
// patch embedding: 1 = tile rows walked sample-fastest so that a pos_embed block is shared by an XCD's whole window (TileIter).
// Measured in round 6 (profiles/r06o_posperm_ab.txt): 2,161 us against 2,098 us per launch at 96 units in storage order -- the
// epilogue does not wait for pos_embed's Infinity-Cache fetches; off
#ifndef SWIFTK_X_POSPERM
#define SWIFTK_X_POSPERM 0
#endif
// SWIGLU_BWD epilogue: R > 0 = a rolling window of R saved pre-activation chunks per lane (R loads in flight) instead of two groups of four.
// Measured in round 6 (profiles/r06m_gemm_ab_bwdroll.txt): R = 8 +3.3 %, R = 10 +4.2 % SLOWER (bit-equal): the epilogue is not short of loads in flight
#ifndef SWIFTK_X_BWDROLL
#define SWIFTK_X_BWDROLL 0
#endif
// cache policy of the bf16 output tiles' 16-B stores: 0 = default, 1 = nt, 2 = sc1 (write-through, line not kept in the
// XCD's L2), 3 = sc0 sc1.  The outputs are written once and never re-read by the kernel; a round of 32 tiles per XCD writes
// 5.8 MB through a 4 MB L2 that should be holding the W panel the XCD re-reads every round.
// SWIGLU_BWD: pull the tile's saved pre-activations (256 rows x 11 lines, 360 KB) into the XCD's L2 during the tile's last
// k-tile (six 4-byte-per-lane LDS-DMA requests per wave into a scratch area), so the epilogue's 16-B reads -- two groups
// of four in flight per lane, twelve groups per tile -- wait for L2 instead of HBM.  Measured 659 against 594 us per launch at
// local batch 8 (profiles/r03p_gemm_ab_hpf.txt): off.
#ifndef SWIFTK_X_HPF
#define SWIFTK_X_HPF 0
#endif
#ifndef SWIFTK_X_STORE
#define SWIFTK_X_STORE 0
#endif
// L2 look-ahead of the persistent kernel (k-tiles): in the second half of every k-tile (where no DMA piece is issued) each
// wave requests one 4-byte LDS-DMA per lane from the 128-B lines its workgroup will stage SWIFTK_X_TOUCH + 1 k-tiles later;
// the k-tile's closing wait leaves those two requests in flight (counted vmcnt), so their miss latency is never waited for.
// non-temporal LDS-DMA for the activation operand (A/B experiment)
#ifndef SWIFTK_X_ANT
#define SWIFTK_X_ANT 0
#endif
#ifndef SWIFTK_X_TOUCH
#define SWIFTK_X_TOUCH 0
#endif
// Ping-pong k-loop (cdna_hip_programming.md section 5, "8-phase" schedule, on this kernel's 256 x 352 geometry): a k-tile is
// four phases (k-half x column half of the wave tile); in every phase a wave first requests the phase's fragments from LDS
// and issues its share of the next stage's DMA pieces (MEM), then -- behind a workgroup barrier -- runs the phase's 20-24 MFMAs
// back to back (COMPUTE), then a second barrier.  Waves 4-7 (the SIMD partners of waves 0-3) run one barrier behind, so on every
// SIMD one wave is in COMPUTE while its partner is in MEM: fragment-read latency and DMA issue never sit between a wave's own
// MFMAs.  1 = per-k-tile drain of the DMA (placed in the last MEM phase); 2 = counted: the W pieces of the second column
// half stay in flight across the k-tile boundary and are waited for in the next k-tile's first MEM phase.
#ifndef SWIFTK_X_PP
#define SWIFTK_X_PP 1
#endif
#ifndef SWIFTK_X_PP_PRIO
#define SWIFTK_X_PP_PRIO 1
#endif
// diagnostic build of the ping-pong loop (never in libswiftk.so): tuning key 3 bits 1 / 4 / 8 as in SWIFTK_GEMM_INSTR (no DMA, no
// epilogue, every stage re-reads k-tile 0), bit 64 = s_memtime stamps of waves 0 and 4 of every 32nd workgroup around every
// barrier of the workgroup's second tile, into the buffer passed as ep1 (EPI_NONE): [wg / 32][wave group][k-tile][16] uint64
#ifndef SWIFTK_PP_STAMP
#define SWIFTK_PP_STAMP 0
#endif
// bf16 epilogue: request all of a 16-row slab's row chunks from LDS before the first store (1) or chunk by chunk (0)
#ifndef SWIFTK_X_EPIBATCH
#define SWIFTK_X_EPIBATCH 0
#endif

struct GemmArgs {
    const char* A;
    const char* W;
    char* C;
    int64_t lda_b, ldw_b;  // row strides in bytes
    int64_t ldc;           // row stride of C in elements
    int M, N, K;
    const float* ep0;
    const float* ep1;
    int pos_rows;
    int ntn;
    int ni;   // W-side MFMA tiles per wave of the persistent kernel's geometry: 10 / 11 / 12 = 320 / 352 / 384 columns per tile
    int qk_only = 0;  // SWIFTK_EPI_QKNORM with fp32 operands: W rows / C columns are [q|k] pairs picked out of [q|k|v] triples (swiftk_gemm, pos_rows < 0)
    int dbg;  // tuning experiments only: 1 = no DMA in the loop, 2 = no barrier (both give wrong results)
    int khalf;         // the last k-tile holds data in its first half only (K = 16.5 tiles for d = 1056)
    int touch;         // persistent kernel: L2 look-ahead requests on (aligned shapes only: M % 256 == 0, N % tile width == 0)
    int stagger;       // persistent kernel: start-up delay step in 10-ns ticks (workgroup phase p waits p x stagger); 0 = off
    int ksplit;        // persistent kernel: k-ranges per output tile (1 = plain)
    int tail_from;     // EPI_NONE_TAIL: tiles >= tail_from (in the walk's tile order) run as two k-halves into slabs 0 / 1 (C + c_split)
    int64_t c_split;   // elements between the fp32 slabs of consecutive splits
    int64_t batch_a, batch_w, batch_c;  // one-tile-per-workgroup kernel only: byte steps of A / W / C per blockIdx.y (batched GEMM)
    // QKNORM only: window-tiled output [sample][window][head][q|k|v][256][88] (t_gw = 0: plain row-major C)
    int t_gh, t_gw, t_sh, t_sw, t_heads;
    // fp32 operands, persistent kernel: two-level accumulation.  Every `kchunk` k-tiles a workgroup parks its accumulators in
    // its private slab of `kscr` (lane-major float4s: blockIdx.x, wave, (i, j), lane) and restarts from zero; a tile's last
    // k-tile adds the parked sum back before the epilogue.  0 = one chain over the whole K range.
    float* kscr;
    int kchunk;
};

template <typename T>
__device__ __forceinline__ void mma_chunk(f32x4& acc, const uint4& wf, const uint4& xf);

template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(f32x4& acc, const uint4& wf, const uint4& xf) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf), acc, 0, 0,
                                                  0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x4& acc, const uint4& wf, const uint4& xf) {
    // hardware k-slot (lane>>4) of MFMA j  <->  actual k = 4*chunk + j, the same bijection for both operands
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf.x), __uint_as_float(xf.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf.y), __uint_as_float(xf.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf.z), __uint_as_float(xf.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf.w), __uint_as_float(xf.w), acc, 0, 0, 0);
}

template <typename OutT>
__device__ __forceinline__ void store4(OutT* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(a, b), pack_bf16(c, d));
}
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
// one 16-B output store with the build's cache policy; inline asm keeps the count of VMEM operations the kernel's
// counted s_waitcnt relies on (the string ends with s_nop 1: the data registers must outlive the issue)
__device__ __forceinline__ void store16_out(void* p, const uint4& q) {
#if SWIFTK_X_STORE == 0
    *reinterpret_cast<uint4*>(p) = q;
#else
    const u32x4 v = {q.x, q.y, q.z, q.w};
#if SWIFTK_X_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif SWIFTK_X_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
#endif
}

template <typename OutT>
__device__ __forceinline__ void store2(OutT* p, float a, float b);
template <>
__device__ __forceinline__ void store2<float>(float* p, float a, float b) {
    *reinterpret_cast<float2*>(p) = make_float2(a, b);
}
template <>
__device__ __forceinline__ void store2<bf16_t>(bf16_t* p, float a, float b) {
    *reinterpret_cast<uint32_t*>(p) = pack_bf16(a, b);
}

// SWIFTK_EPI_QKNORM: the wave's 64 x 176 tile of to_qkv's output is exactly two 88-wide head vectors per row
// (columns [c0, c0+88) and [c0+88, c0+176); vector index v = column/88 -> head v/3, kind v%3 = q|k|v).  Cosine
// attention's prologue (swinv2.py:123-127) happens here on the fp32 accumulators: q <- q/max(|q|,1e-12) *
// exp(min(scale_h, ln 100)), k <- k/max(|k|,1e-12), v untouched.  A row's 88 values sit in the four 16-lane
// groups of the wave: 22 accumulator quads -> register sums + two cross-group shuffles.
// `rn` (optional, training): 1/max(|.|, 1e-12) of every q / k vector, [M][N/88] fp32 (1 for v), for the backward pass.
template <int NI>
__device__ __forceinline__ void qknorm_tile(f32x4 (&acc)[MI][NI], int lane, int c0, const float* __restrict__ scale,
                                            float* __restrict__ rn = nullptr, int mrow0 = 0, int M = 0, int nvec = 0, bool qk_only = false) {
    constexpr int HD = 8 * NI;  // the wave tile's 16*NI columns are two head vectors: 80 / 88 / 96 for NI = 10 / 11 / 12
    const int g4 = lane >> 4;
    const int vA = c0 / HD, vB = vA + 1;
    // (qk_only: the columns are [q | k] pairs -- vector v is q or k of head v / 2 -- instead of [q | k | v] triples)
    const int kA = qk_only ? (vA & 1) : vA % 3, kB = qk_only ? (vB & 1) : vB % 3;
    // (nvec > 0: vectors at or beyond it lie outside the matrix -- N = 3 head_dim, ONE head, leaves the tile's fourth vector empty --
    // and must not read a logit scale: their accumulators are never stored)
    const bool inA = nvec <= 0 || vA < nvec, inB = nvec <= 0 || vB < nvec;
    const float tauA = kA == 0 && inA ? expf(fminf(scale[qk_only ? vA >> 1 : vA / 3], 4.605170185988092f)) : 1.0f;
    const float tauB = kB == 0 && inB ? expf(fminf(scale[qk_only ? vB >> 1 : vB / 3], 4.605170185988092f)) : 1.0f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const f32x4 v = acc[i][j];
            const float t = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            // quad (j, g4) holds columns 16 j + 4 g4 .. + 3: first vector below HD, second from HD on (for HD = 88 the
            // boundary runs through j = 5 between lane groups 1 and 2)
            if (16 * j + 12 < HD) sa += t;
            else if (16 * j >= HD) sb += t;
            else { sa += 16 * j + 4 * g4 < HD ? t : 0.f; sb += 16 * j + 4 * g4 < HD ? 0.f : t; }
        }
        sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
        sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
        const float fa = kA == 2 ? 1.0f : tauA / fmaxf(sqrtf(sa), 1e-12f);
        const float fb = kB == 2 ? 1.0f : tauB / fmaxf(sqrtf(sb), 1e-12f);
        if (rn && g4 == 0) {
            const int m = mrow0 + i * 16 + (lane & 15);
            if (m < M) {
                if (inA) rn[(int64_t)m * nvec + vA] = kA == 2 ? 1.0f : 1.0f / fmaxf(sqrtf(sa), 1e-12f);
                if (inB) rn[(int64_t)m * nvec + vB] = kB == 2 ? 1.0f : 1.0f / fmaxf(sqrtf(sb), 1e-12f);
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float f = 16 * j + 12 < HD ? fa : (16 * j >= HD ? fb : (16 * j + 4 * g4 < HD ? fa : fb));
            acc[i][j] *= f;
        }
    }
}

// SWIFTK_EPI_QKNORM_JVP: the same prologue AND its tangent (swinv2.py:123-127 under torch.func.jvp) on a PAIRED wave tile --
// accumulator rows i = 0, 1 are primal rows, i = 2, 3 the tangent rows of the same tokens (gemm_kernel_p, PAIRED), so a lane
// holds v and dv of one element:  v-hat = tau v / n,  d v-hat = tau / n (dv - v (v . dv) / n^2),  n = max(|v|, 1e-12).
// This half computes the row statistics: per row block i and head vector (A | B) the factors f = tau / n and f c = f (v . dv) / n^2,
// so that v-hat = f v and d v-hat = f dv - (f c) v are formed where the tile is packed for its stores (formed here, the 88 scaled
// quads sit in fresh registers until the store loop takes them and the k-loop's invariants spill).
template <int NI>
__device__ __forceinline__ void qknorm_jvp_stats(const f32x4 (&acc)[MI][NI], int lane, int c0, const float* __restrict__ scale,
                                                 float* __restrict__ rn, int mrow0, int nvec, float (&qf)[MI / 2][4]) {
    constexpr int HD = 8 * NI;
    const int g4 = lane >> 4;
    const int vA = c0 / HD, vB = vA + 1;
    const int kA = vA % 3, kB = vB % 3;
    const float tauA = kA == 0 ? expf(fminf(scale[vA / 3], 4.605170185988092f)) : 1.0f;
    const float tauB = kB == 0 ? expf(fminf(scale[vB / 3], 4.605170185988092f)) : 1.0f;
#pragma unroll
    for (int i = 0; i < MI / 2; ++i) {
        float sa = 0.f, sb = 0.f, da = 0.f, db = 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const f32x4 v = acc[i][j], t = acc[i + 2][j];
            if (16 * j + 12 < HD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { sa = fmaf(v[e], v[e], sa); da = fmaf(v[e], t[e], da); }
            } else if (16 * j >= HD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { sb = fmaf(v[e], v[e], sb); db = fmaf(v[e], t[e], db); }
            } else {  // (for HD = 88 the boundary runs through j = 5 between lane groups 1 and 2)
                const bool first = 16 * j + 4 * g4 < HD;
                float ss = 0.f, dd = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ss = fmaf(v[e], v[e], ss); dd = fmaf(v[e], t[e], dd); }
                sa += first ? ss : 0.f; da += first ? dd : 0.f;
                sb += first ? 0.f : ss; db += first ? 0.f : dd;
            }
        }
        sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
        sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
        da += __shfl_xor(da, 16, 64); da += __shfl_xor(da, 32, 64);
        db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
        // one division per vector (1 / n), the rest are products of it
        const float ra = kA == 2 ? 1.0f : 1.0f / fmaxf(sqrtf(sa), 1e-12f), rb = kB == 2 ? 1.0f : 1.0f / fmaxf(sqrtf(sb), 1e-12f);
        qf[i][0] = kA == 2 ? 1.0f : tauA * ra;
        qf[i][1] = kB == 2 ? 1.0f : tauB * rb;
        qf[i][2] = kA == 2 ? 0.0f : qf[i][0] * da * ra * ra;
        qf[i][3] = kB == 2 ? 0.0f : qf[i][1] * db * rb * rb;
        if (rn && g4 == 0 && vB < nvec) {
            const int m = mrow0 + i * 16 + (lane & 15);
            rn[(int64_t)m * nvec + vA] = ra;
            rn[(int64_t)m * nvec + vB] = rb;
        }
    }
}

template <typename T, typename OutT, int EPI>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
    // Two separate LDS objects (not one array carved in two): hipcc tags accesses to distinct LDS variables with
    // distinct alias scopes, and only then does its waitcnt pass let a ds_read of one buffer proceed while the
    // LDS-DMA into the other is still in flight (otherwise it drains vmcnt(0) before every fragment read).
    __shared__ __attribute__((aligned(16))) char stage0[STAGE];
    __shared__ __attribute__((aligned(16))) char stage1[STAGE];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 1, wn = wv & 1;

    // XCD-aware, bijective block -> tile map: blocks with equal id%8 share an XCD (and its L2); give each
    // XCD a contiguous run of tiles, N fastest, so the A panel of a tile row is fetched into one L2 only.
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (tile / g.ntn) * BM;
    const int n0 = (tile % g.ntn) * BN;
    // batched form (swiftk_gemm_batched): matrix blockIdx.y of a stack with constant steps
    g.A += (int64_t)blockIdx.y * g.batch_a;
    g.W += (int64_t)blockIdx.y * g.batch_w;
    g.C += (int64_t)blockIdx.y * g.batch_c;

    // ---- per-lane source pointers of the LDS-DMA pieces (k-tile 0) ----
    const int prow = lane >> 3;      // row inside an 8-row piece
    const int pchunk = lane & 7;     // physical 16-B chunk this lane's bytes land in
    const char* asrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wv * 4 + i) * 8 + prow;
        const int logical = pchunk ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < g.M ? gm : g.M - 1;
        asrc[i] = g.A + (int64_t)gm * g.lda_b + logical * 16;
    }
    const char* bsrc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = (wv + 8 * i) * 8 + prow;
        const int logical = pchunk ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < g.N ? gn : g.N - 1;
        bsrc[i] = g.W + (int64_t)gn * g.ldw_b + logical * 16;
    }

    auto stage_load = [&](char* sa, int kt) {
        char* sb = sa + A_BYTES;
        const int64_t koff = (int64_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(asrc[i] + koff), LDS_PTR(sa + (wv * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i < 5 || wv < 4)
                __builtin_amdgcn_global_load_lds(GLB_PTR(bsrc[i] + koff), LDS_PTR(sb + (wv + 8 * i) * 1024), 16, 0, 0);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row r16 of a 16-row MFMA tile, logical chunk (lane>>4) + 4*ks
    const int r16 = lane & 15;
    const int xoff = (wm * 64 + r16) * ROWB;
    const int woff = A_BYTES + (wn * 176 + r16) * ROWB;
    const int ch0 = (((lane >> 4) + 0) ^ (r16 >> 1)) * 16;
    const int ch1 = (((lane >> 4) + 4) ^ (r16 >> 1)) * 16;

    auto compute = [&](const char* s) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ks ? ch1 : ch0;
            uint4 xf[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const uint4*>(s + xoff + i * 16 * ROWB + ch);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const uint4 wf = *reinterpret_cast<const uint4*>(s + woff + j * 16 * ROWB + ch);
#pragma unroll
                for (int i = 0; i < MI; ++i) mma_chunk<T>(acc[i][j], wf, xf[i]);
            }
        }
    };

    const int nk = g.K / (ROWB / (int)sizeof(T));
    stage_load(stage0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        // own DMA of tile kt has landed (vmcnt(0)); every wave is done reading the other buffer (barrier)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (kt + 1 < nk) stage_load(stage1, kt + 1);
        compute(stage0);
        if (kt + 1 < nk) {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            if (kt + 2 < nk) stage_load(stage0, kt + 2);
            compute(stage1);
        }
    }

    // ---- epilogue: lane holds C[m][nb .. nb+3] for m = ..+r16, nb = ..+4*(lane>>4) ----
    if constexpr (EPI == SWIFTK_EPI_QKNORM || EPI == EPI_QKNORM_TILED)
        qknorm_tile<NI>(acc, lane, n0 + wn * 176, g.ep0, const_cast<float*>(g.ep1), m0 + wm * 64, g.M, g.N / 88);
    OutT* C = reinterpret_cast<OutT*>(g.C);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * 64 + i * 16 + r16;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nb = n0 + wn * 176 + j * 16 + 4 * (lane >> 4);
            if (nb >= g.N) continue;
            f32x4 v = acc[i][j];
            if constexpr (EPI == SWIFTK_EPI_BIAS_POS) {
                const float4 b = *reinterpret_cast<const float4*>(g.ep0 + nb);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                if (g.ep1) {
                    const float4 p = *reinterpret_cast<const float4*>(g.ep1 + (int64_t)(m % g.pos_rows) * g.N + nb);
                    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
                }
            }
            if constexpr (EPI == SWIFTK_EPI_SWIGLU) {
                const float h0 = swiglu_out<OutT>(v[0], v[1]);
                const float h1 = swiglu_out<OutT>(v[2], v[3]);
                store2<OutT>(C + (int64_t)m * g.ldc + (nb >> 1), h0, h1);
            } else {
                store4<OutT>(C + (int64_t)m * g.ldc + nb, v[0], v[1], v[2], v[3]);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Persistent variant: a fixed grid of workgroups walks the tile list, and the two-stage LDS-DMA pipeline runs
// straight through tile boundaries -- the first k-tile of the next output tile is already in flight while the
// epilogue of the current one stores, so there is no per-tile prologue bubble.  Tile order is grouped (GM tile
// rows x all tile columns, column-major inside a group) so that the ~32 tiles an XCD works on at any moment
// form an 8 x 4 block sharing A panels and W panels in that XCD's L2; the DMA pieces of the next stage are
// issued between MFMA groups instead of in one burst after the barrier.
struct TileIter {
    int ntm, ntn, gm;  // tile rows, tile cols, group height
    int pb, pt;        // > 1: tile rows are walked sample-fastest (pb samples of pt tile rows each), see below
    __device__ __forceinline__ void coords(int t, int& tm, int& tn) const {
        const int per = gm * ntn;
        const int grp = t / per, r = t - grp * per;
        const int rows = min(gm, ntm - grp * gm);
        tn = r / rows;
        tm = grp * gm + (r - tn * rows);
        // (SWIFTK_X_POSPERM experiment, off: patch embedding with the tile rows walked sample-fastest -- an XCD's window then covers
        // the SAME token block of eight samples, so a pos_embed block is fetched once per window instead of once per sample)
        if (pb > 1) tm = (tm % pb) * pt + tm / pb;
    }
};

template <typename T, typename OutT, int EPI, int NI, bool PPK>
__global__ __launch_bounds__(NT) void gemm_kernel_p(GemmArgs g, int ntm, int gm) {
    // geometry of this instantiation: 8 waves as 4 (M) x 2 (N), each 64 x 16 NI
    constexpr int WT = 16 * NI;                 // columns of a wave tile (160 / 176 / 192)
    constexpr int BN = 2 * WT;                  // 320 / 352 / 384
    constexpr int B_BYTES = BN * ROWB;          // 40 / 44 / 48 KiB
    constexpr int STAGE = A_BYTES + B_BYTES;    // 72 / 76 / 80 KiB: two stages = 144 / 152 / 160 KiB of the CU's 160
    constexpr int WP = B_BYTES / 1024;          // W pieces per stage: 40 / 44 / 48 = 5, 5.5, 6 per wave
    constexpr int HD = 8 * NI;                  // QKNORM: head_dim (a wave tile = two head vectors)
    constexpr bool TOUCH = SWIFTK_X_TOUCH > 0 && NI <= 11 && sizeof(T) == 2;  // (384-wide tiles use all 160 KiB of LDS)
    constexpr bool HPF = SWIFTK_X_HPF > 0 && EPI == SWIFTK_EPI_SWIGLU_BWD && NI <= 11 && sizeof(T) == 2;
    // PAIRED (swiftk_gemm_jvp): A holds primal rows 0..M/2-1 and tangent rows M/2..M-1; tile tm takes primal rows 128 tm .. + 127
    // and their tangent rows, wave wm the 32 + 32 rows of tokens 128 tm + 32 wm .. + 31: accumulator row blocks i = 0, 1 are primal,
    // i = 2, 3 the tangents of the same tokens -- the epilogue's tangent rules find both values of an element in one lane
    constexpr bool PAIRED = EPI == SWIFTK_EPI_QKNORM_JVP || EPI == SWIFTK_EPI_SWIGLU_JVP;
    // PPK: the ping-pong k-loop (needs at least three k-tiles per work item; the launcher checks)
    constexpr bool PP = PPK && sizeof(T) == 2 && !TOUCH && !HPF && !SWIFTK_GEMM_INSTR;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + (TOUCH || HPF ? 256 : 0)];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 1, wn = wv & 1;
#if SWIFTK_X_POSPERM
    const bool posperm = EPI == EPI_BIAS_POS_PAIR && g.ep1 && g.pos_rows >= BM && g.pos_rows % BM == 0 && g.M % g.pos_rows == 0 && g.ksplit == 1;
    const TileIter it{ntm, g.ntn, gm, posperm ? (int)(g.M / g.pos_rows) : 0, posperm ? (int)(g.pos_rows / BM) : 0};
#else
    const TileIter it{ntm, g.ntn, gm, 0, 0};
#endif
    // work item = (output tile, k-split): with ksplit > 1 (weight gradients: few output tiles, K = all tokens) each
    // split accumulates its k-range into its own fp32 slab C + split*c_split; a reduce kernel sums the slabs
    // EPI_NONE_TAIL (small batches, ksplit = 1): work items of two sizes -- tiles [0, tail_from) whole, then each later tile as two
    // k-halves, split s into slab s.  tail_from is a multiple of the grid size and at most half a round of tiles follows it, so the
    // round-robin walk below hands every workgroup its whole tiles and then at most ONE half: the last round costs half a tile time
    constexpr bool TAIL = EPI == EPI_NONE_TAIL;
    const int ksplit = TAIL ? 1 : g.ksplit;
    const int tail_from = TAIL ? g.tail_from : 0;
    const int ntiles = TAIL ? 2 * ntm * g.ntn - tail_from : ntm * g.ntn * ksplit;
    auto tile_of = [&](int item) {
        if constexpr (TAIL) return item < tail_from ? item : tail_from + ((item - tail_from) >> 1);
        else return item / ksplit;
    };
    auto split_of = [&](int item) {
        if constexpr (TAIL) return item < tail_from ? 0 : (item - tail_from) & 1;
        else return item % ksplit;
    };
    // workgroups with equal blockIdx%8 share an XCD: give each XCD a contiguous run of virtual ids
    int vid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int stride = gridDim.x;
    if (vid >= ntiles) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

    // LDS-DMA addressing: per piece a wave-uniform 64-bit base (SGPRs: tile origin + piece row + k offset) plus a
    // per-lane 32-bit offset (row-in-piece * ld + swizzled 16-B chunk).  The swizzle term (row>>1)&7 depends on the
    // piece only through its parity for A (pieces 4*wv+i) and not at all for W (pieces wv+8i), so three VGPRs serve
    // all ten pieces.  Needs M % 8 == 0 and N % 8 == 0 (a piece is entirely inside or outside the matrix).
    const int prow = lane >> 3, pchunk = lane & 7;
    const uint32_t va_even = (uint32_t)(prow * g.lda_b) + 16u * (pchunk ^ ((prow >> 1) & 7));
    const uint32_t va_odd = (uint32_t)(prow * g.lda_b) + 16u * (pchunk ^ ((4 + (prow >> 1)) & 7));
    const uint32_t vb = (uint32_t)(prow * g.ldw_b) + 16u * (pchunk ^ ((4 * (wv & 1) + (prow >> 1)) & 7));
    // L2 look-ahead: the tile's 256 + BN operand rows as one list, 76 rows per wave: request 1 = rows 64 wv .. + 63 of the list
    // (waves 0-3: A rows, waves 4-7: W rows 0..255), request 2 = W rows 256 + 12 wv .. + 11 (lanes 12.. repeat the last one)
    // (row numbers relative to the tile; clamped against the matrix edge where the request is built)
    const int tr1 = (wv & 3) * 64 + lane;
    const int tr2 = min(256 + wv * ((BN - 256 + 7) / 8) + min(lane, (BN - 256 + 7) / 8 - 1), BN - 1);
    int t_m0 = 0, t_n0 = 0;  // origin of the tile the DMA currently feeds (set_sources)
    const int nk_all = g.K / (ROWB / (int)sizeof(T));
    auto k_begin = [&](int item) {
        if constexpr (TAIL) return item >= tail_from && ((item - tail_from) & 1) ? nk_all >> 1 : 0;
        else return (int)((int64_t)(item % ksplit) * nk_all / ksplit);
    };
    auto k_end = [&](int item) {
        if constexpr (TAIL) return item >= tail_from && !((item - tail_from) & 1) ? nk_all >> 1 : nk_all;
        else return (int)((int64_t)(item % ksplit + 1) * nk_all / ksplit);
    };
    int tile = vid, kt = k_begin(vid);
    // Row bases of this wave's ten pieces for the tile the DMA currently feeds: computed once per tile and kept in
    // SGPRs, so issuing a piece costs three instructions (M0, nop, load) instead of ~20 scalar address ops -- at
    // ten pieces per k-tile the scalar arithmetic alone used to take as many issue slots as the 88 MFMAs.
    const char* abase[4];
    const char* wbase[6];
    auto set_sources = [&](int t) {
        int tm, tn;
        it.coords(tile_of(t), tm, tn);
        t_m0 = tm * BM;
        t_n0 = tn * BN;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int rb = tm * BM + (wv * 4 + p) * 8;
            if constexpr (PAIRED)  // piece (wv, p) = tile rows 64 (wv >> 1) + 32 (wv & 1) + 8 p ..: row block i = 2 (wv & 1) + (p >> 1)
                rb = ((wv & 1) ? (g.M >> 1) : 0) + tm * (BM / 2) + (wv >> 1) * 32 + (p >> 1) * 16 + (p & 1) * 8;
            rb = rb < g.M ? rb : g.M - 8;
            abase[p] = g.A + (int64_t)rb * g.lda_b;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int rb = tn * BN + (wv + 8 * i) * 8;
            rb = rb < g.N ? rb : g.N - 8;
            if constexpr (sizeof(T) == 4 && EPI == SWIFTK_EPI_QKNORM)  // [q | k] pairs out of to_qkv's [q | k | v] row triples: skip the v rows
                if (g.qk_only) rb += (rb / (2 * HD)) * HD;
            wbase[i] = g.W + (int64_t)rb * g.ldw_b;
        }
    };
    // Ping-pong loop: the W rows of a stage form two regions with lives of their own -- W0 = the first JA column blocks of both
    // wave-tile halves (read in a k-tile's first two phases), W1 = the other JB (last two phases).  W0 pieces n = wv + 8 i < 4 JA,
    // W1 pieces n = wv + 8 i < 4 JB; piece n of a region: half h = n / (2 J), rows h WT + [JA 16 +] 8 (n - 2 J h) .. + 7 of the
    // tile (2 J and WT / 8 are even: the swizzle parity of a piece is wv & 1, so `vb` serves every W piece here too).
#ifndef SWIFTK_PP_JA_HI
#define SWIFTK_PP_JA_HI 0
#endif
    // (k-half major order: the LOW column half is the smaller one -- its MEM phase carries the four activation fragments too and
    // runs beside the partner's MFMAs of the high half, so the longer MEM phase meets the longer COMPUTE phase)
    constexpr int JA = SWIFTK_PP_JA_HI ? (NI + 1) / 2 : NI / 2, JB = NI - JA, JM = JA > JB ? JA : JB;
    const char* w0base[3];
    const char* w1base[3];
    auto w0row = [&](int i) { const int n = wv + 8 * i, h = n >= 2 * JA; return h * WT + (n - 2 * JA * h) * 8; };
    auto w1row = [&](int i) { const int n = wv + 8 * i, h = n >= 2 * JB; return h * WT + JA * 16 + (n - 2 * JB * h) * 8; };
    auto set_sources_aw1 = [&](int t) {
        int tm, tn;
        it.coords(tile_of(t), tm, tn);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int rb = tm * BM + (wv * 4 + p) * 8;
            if constexpr (PAIRED) rb = ((wv & 1) ? (g.M >> 1) : 0) + tm * (BM / 2) + (wv >> 1) * 32 + (p >> 1) * 16 + (p & 1) * 8;
            rb = rb < g.M ? rb : g.M - 8;
            abase[p] = g.A + (int64_t)rb * g.lda_b;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int rb = tn * BN + w1row(i);
            rb = rb < g.N ? rb : g.N - 8;
            w1base[i] = g.W + (int64_t)rb * g.ldw_b;
        }
    };
    auto set_sources_w0 = [&](int t) {
        int tm, tn;
        it.coords(tile_of(t), tm, tn);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int rb = tn * BN + w0row(i);
            rb = rb < g.N ? rb : g.N - 8;
            w0base[i] = g.W + (int64_t)rb * g.ldw_b;
        }
    };
    auto pp_a = [&](uint32_t sa, uint32_t koff, int p) {
        dma_piece_fast(sa + (wv * 4 + p) * 1024, abase[p], ((p & 1) ? va_odd : va_even) + koff);
    };
    auto pp_w0 = [&](uint32_t sa, uint32_t koff, int i) {
        if (wv + 8 * i < 4 * JA) dma_piece_fast(sa + A_BYTES + w0row(i) * ROWB, w0base[i], vb + koff);
    };
    auto pp_w1 = [&](uint32_t sa, uint32_t koff, int i) {
        if (wv + 8 * i < 4 * JB) dma_piece_fast(sa + A_BYTES + w1row(i) * ROWB, w1base[i], vb + koff);
    };
    // piece p of the stage at LDS byte address `sa`: pieces 0-3 = A rows, 4-9 = W rows (1 KiB = 8 rows x 128 B);
    // `koff` = byte offset of the k-tile inside a row, carried in the per-lane offset
    auto issue_piece = [&](uint32_t sa, uint32_t koff, int p) {
        if (p < 4) {
#if SWIFTK_X_ANT
            dma_piece_fast_nt(sa + (wv * 4 + p) * 1024, abase[p], ((p & 1) ? va_odd : va_even) + koff);
#else
            dma_piece_fast(sa + (wv * 4 + p) * 1024, abase[p], ((p & 1) ? va_odd : va_even) + koff);
#endif
        } else {
            const int i = p - 4;
            // W has 40 / 44 / 48 pieces for 8 waves: with 44 the sixth one exists for waves 0-3 only (a wave-uniform branch
            // around the asm; the EXEC-mask form needs an SGPR operand and hipcc hands inline asm a VGPR for it at this
            // SGPR pressure)
            if (wv + 8 * i < WP) dma_piece_fast(sa + A_BYTES + (wv + 8 * i) * 1024, wbase[i], vb + koff);
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15;
    const int xoff = (wm * 64 + r16) * ROWB;
    const int woff = A_BYTES + (wn * WT + r16) * ROWB;
    const int ch0 = (((lane >> 4) + 0) ^ (r16 >> 1)) * 16;
    const int ch1 = (((lane >> 4) + 4) ^ (r16 >> 1)) * 16;

    // Flattened (tile, k-tile) walk.  Each step computes from one stage while the DMA of the following step fills
    // the other; past the very last step the "following step" is a harmless re-load of this tile's first k-tile.
    int nk = k_end(tile);
    if constexpr (PP) {
        set_sources_aw1(tile);
        set_sources_w0(tile);
#pragma unroll
        for (int p = 0; p < 4; ++p) pp_a(lds0, (uint32_t)kt * ROWB, p);
#pragma unroll
        for (int i = 0; i < 3; ++i) pp_w0(lds0, (uint32_t)kt * ROWB, i);
#pragma unroll
        for (int i = 0; i < 3; ++i) pp_w1(lds0, (uint32_t)kt * ROWB, i);
    } else {
        set_sources(tile);
#pragma unroll
        for (int p = 0; p < 10; ++p) issue_piece(lds0, (uint32_t)kt * ROWB, p);
    }
    int par = 0;
    bool have_part = false;  // (fp32 operands, kchunk > 0) this tile has a partial sum parked in the workgroup's scratch slab
    // dbg bit 32 (timing experiment, EPI_NONE only): wave 0 of every 32nd workgroup logs s_memtime at five points of each tile
    // into the buffer passed as ep1 -- [wg/32][tile][8] uint64: loop top of the first k-step, last MFMA issued, epilogue
    // barrier passed, stores issued, next loop top passed, sum of the k-steps' vmcnt(0) waits, sum of their barrier waits
#if SWIFTK_GEMM_INSTR
    unsigned long long* tlog = nullptr;
    int tl_i = 0;
    if ((g.dbg & 32) && wv == 0 && lane == 0 && (blockIdx.x & 31) == 0)
        tlog = reinterpret_cast<unsigned long long*>(const_cast<float*>(g.ep1)) + (blockIdx.x >> 5) * 8 * 64;
    bool first_k = true;
#endif
    // Start-up stagger.  All 256 workgroups walk equally long tiles from the same instant, so every epilogue -- 180 KB of
    // stores per workgroup -- falls into the same few microseconds: the chip alternates between an HBM-write burst with
    // idle matrix pipes (6-8 us per tile at 96 units, 6 TB/s) and a k-loop with an idle write path.  Delaying the
    // workgroups by eighths of a tile time (phase = tile row + tile column inside the XCD's 8 x 4 window: the four
    // workgroups sharing an A panel stay within half a tile of each other, the eight sharing a W panel cover all phases)
    // spreads the epilogues over the whole tile period for the rest of the launch.
    // (negative step: phase = XCD number -- the XCDs share no L2, so nothing pulls them back into step once they are apart,
    // while the 32 workgroups of one XCD stay together on their shared panels)
    if (g.stagger != 0) {
        const int ph = g.stagger > 0 ? (((vid % gm) + (vid / gm)) & 7) : (int)(blockIdx.x & 7);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        const uint64_t wait = (uint64_t)ph * (uint64_t)(g.stagger > 0 ? g.stagger : -g.stagger);
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
#if SWIFTK_X_PRIO
    // the second-dispatched half of the workgroup loses every issue arbitration against its SIMD partner; one static
    // priority for that half, no per-segment flips (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (!PP && wv >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int grp = wv >> 2;  // ping-pong: waves wv and wv + 4 share a SIMD; group 1 runs one barrier behind group 0
    bool first_kt = true;     // ping-pong: this trip is the first k-tile of an output tile (all waves aligned at its top)
#if SWIFTK_PP_STAMP
    unsigned long long* plog = nullptr;
    int pl_tile = 0, pl_kt = 0, pl_i = 0;
    if ((g.dbg & 64) && (wv & 3) == 0 && lane == 0 && (blockIdx.x & 31) == 0)
        plog = reinterpret_cast<unsigned long long*>(const_cast<float*>(g.ep1)) + ((blockIdx.x >> 5) * 2 + (wv >> 2)) * 20 * 16;
#define PP_STAMP() do { if (plog && pl_tile == 1 && pl_i < 16) plog[pl_kt * 16 + pl_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_STAMP() do {} while (0)
#endif
    for (;;) {
        // (waited at the bottom of the previous trip) own DMA of the stage about to be read has landed; after the
        // barrier every wave's has, and every wave is done reading the other stage (its fragment reads were consumed
        // by MFMAs before it got here)
#if SWIFTK_GEMM_INSTR
        unsigned long long ts0 = 0, ts1 = 0;
        if (tlog) ts0 = ts1 = __builtin_amdgcn_s_memtime();
#endif
#if SWIFTK_GEMM_INSTR
        if (tlog) ts1 = __builtin_amdgcn_s_memtime();
        if (!(g.dbg & 2)) __builtin_amdgcn_s_barrier();
        if (tlog) {
            const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
            if (first_k) {
                if (tl_i > 0) tlog[(tl_i - 1) * 8 + 4] = ts2;
                tlog[tl_i * 8 + 0] = ts2;
                tlog[tl_i * 8 + 5] = 0;
                tlog[tl_i * 8 + 6] = 0;
            } else {  // waits of the k-steps inside the tile: own DMA, then the other waves
                tlog[tl_i * 8 + 5] += ts1 - ts0;
                tlog[tl_i * 8 + 6] += ts2 - ts1;
            }
        }
        first_k = false;
#else
        if constexpr (PP) {
            // only a tile's first k-tile meets all waves aligned (the stage about to be read was waited for by every wave in front
            // of this barrier); inside a tile the phase barriers hand the stages over.  Group 1 then drops one barrier behind.
            if (first_kt) {
                __builtin_amdgcn_s_barrier();
                if (grp) __builtin_amdgcn_s_barrier();
            }
        } else {
            __builtin_amdgcn_s_barrier();
        }
#endif
        const char* s = smem + par * STAGE;
        const uint32_t fill = lds0 + (par ^ 1) * STAGE;
        const bool last_k = (kt + 1 == nk);
        const bool half = g.khalf && (kt + 1 == nk_all);
        uint32_t koff = (uint32_t)(kt + 1) * ROWB;
#if SWIFTK_GEMM_INSTR
        if (g.dbg & 8) koff = 0;  // timing experiment: every stage re-reads k-tile 0 (L2-resident after a tile's first trip)
#endif
        const char* hpf_base = nullptr;  // SWIGLU_BWD look-ahead: this wave's 32 rows of the tile's saved pre-activations
        int hpf_rows = 0, hpf_cols = 0;
        if constexpr (HPF) {
            if (last_k) {  // (t_m0 / t_n0 still name the tile being computed: set_sources(next tile) comes below)
                const int rb = min(t_m0 + wv * 32, (int)g.M - 1);
                hpf_rows = min(31, (int)g.M - 1 - rb);
                hpf_cols = (int)g.N * 4 - 4 - t_n0 * 4;  // last byte offset a request may start at, relative to the tile's first column
                hpf_base = reinterpret_cast<const char*>(g.ep1) + ((int64_t)rb * g.pos_rows + 2 * t_n0) * 2;
            }
        }
        if constexpr (PP) {
            const int ntile = tile + stride < ntiles ? tile + stride : tile;  // (past the last item: harmless re-loads)
            if (last_k) {
                set_sources_aw1(ntile);
                set_sources_w0(ntile);
                koff = (uint32_t)k_begin(ntile) * ROWB;
            }
        } else
        if (last_k) {
            const int ntile = tile + stride;
            if (ntile < ntiles) set_sources(ntile);
            koff = (uint32_t)k_begin(ntile < ntiles ? ntile : tile) * ROWB;
        }
        // One k-tile = 22 steps of 4 MFMAs (one W fragment x four activation fragments), written as straight-line code:
        // the fragment of step i+1 is requested before the MFMAs of step i issue, one DMA piece of the next stage follows
        // each of the first ten steps, and the only control flow is ONE branch around the second k-half (K = 1056 is 16.5
        // k-tiles: the last k-tile of the K range carries data in its first half only; its second half meets zero pad
        // columns, so those 44 MFMAs are skipped -- 3 % of the GEMM).  Branches between the MFMA groups make hipcc
        // shuffle and spill accumulators (297 v_mov + 27 spilled dwords per k-tile in the SwiGLU build).
        if constexpr (PP) {
            // Ping-pong form of the same k-tile ("The k-loop"): phases (k-half, column part) = (0, lo) (0, hi)
            // (1, lo) (1, hi); the activation fragments are read once per k-half.  The last MEM phase ends with lgkmcnt(0) in front
            // of its barrier: no fragment read is pending when the partner group, one barrier later, lets DMA into the stage.
            constexpr int C1L = 3, C1H = 4 * JB >= 24 ? 3 : 2;  // W pieces of the high column part per wave (waves 0-3 | 4-7)
            uint4 xf[MI], wf[JM];
#if SWIFTK_PP_STAMP
            pl_i = 0;
            if (g.dbg & 8) koff = 0;
#define PP_DMA(x) do { if (!(g.dbg & 1)) { x; } } while (0)
#else
#define PP_DMA(x) do { x; } while (0)
#endif
            auto bar2 = [&] {
                __builtin_amdgcn_sched_barrier(0);
                PP_STAMP();
                __builtin_amdgcn_s_barrier();
                PP_STAMP();
                __builtin_amdgcn_sched_barrier(0);
            };
            auto comp = [&](const int j0, const int nj) {
#if SWIFTK_X_PP_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int jj = 0; jj < JM; ++jj) {
                    if (jj < nj) {
#pragma unroll
                        for (int i = 0; i < MI; ++i) mma_chunk<T>(acc[i][j0 + jj], wf[jj], xf[i]);
                    }
                }
#if SWIFTK_X_PP_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            };
            // k-half major: (0, lo) (0, hi) (1, lo) (1, hi); activation fragments are read once per k-half.  Issue order of a
            // wave per k-tile, everything into the other stage:  MEM 0: A0 A1 A2, wait W1(this k-tile) | MEM 1: A3 W0a W0b |
            // MEM 2: W0c W1a | MEM 3: W1b W1c, wait A, W0(next k-tile).  The second wait leaves the C1 youngest requests (W1)
            // in flight across the k-tile boundary; the first one -- three requests later -- retires them.
            auto rdx = [&](const int ch) {
#pragma unroll
                for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const uint4*>(s + xoff + i * 16 * ROWB + ch);
            };
            auto rdw = [&](const int ch, const int j0, const int nj) {
#pragma unroll
                for (int jj = 0; jj < JM; ++jj)
                    if (jj < nj) wf[jj] = *reinterpret_cast<const uint4*>(s + woff + (j0 + jj) * 16 * ROWB + ch);
            };
            // ---- (0, lo)
            rdx(ch0);
            rdw(ch0, 0, JA);
            PP_DMA(pp_a(fill, koff, 0));
            PP_DMA(pp_a(fill, koff, 1));
            PP_DMA(pp_a(fill, koff, 2));
            if (!first_kt) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            bar2();
            comp(0, JA);
            bar2();
            // ---- (0, hi)
            rdw(ch0, JA, JB);
            PP_DMA(pp_a(fill, koff, 3));
            PP_DMA(pp_w0(fill, koff, 0));
            PP_DMA(pp_w0(fill, koff, 1));
            if (half) {  // the k-tile ends here: the rest of its requests, and its last fragment reads
                PP_DMA(pp_w0(fill, koff, 2));
                PP_DMA(pp_w1(fill, koff, 0));
                PP_DMA(pp_w1(fill, koff, 1));
                PP_DMA(pp_w1(fill, koff, 2));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            bar2();
            comp(JA, JB);
            bar2();
            if (!half) {
                // ---- (1, lo)
                rdx(ch1);
                rdw(ch1, 0, JA);
                PP_DMA(pp_w0(fill, koff, 2));
                PP_DMA(pp_w1(fill, koff, 0));
                bar2();
                comp(0, JA);
                bar2();
                // ---- (1, hi)
                rdw(ch1, JA, JB);
                PP_DMA(pp_w1(fill, koff, 1));
                PP_DMA(pp_w1(fill, koff, 2));
                if (!last_k) {
                    if (wv < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C1L) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C1H) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bar2();
                comp(JA, JB);
                bar2();
            }
#undef PP_DMA
        } else {
            uint4 xf[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const uint4*>(s + xoff + i * 16 * ROWB + ch0);
            // look-ahead target: k-tile kt + 1 + SWIFTK_X_TOUCH of the tile the DMA feeds (its last one at most); in a tile's last
            // k-tile the DMA already feeds the next tile (set_sources above): the second k-tile of that one
            uint32_t toff = 0;
            if constexpr (TOUCH) {
                const int tk = last_k ? (int)(koff / ROWB) + 1 : min(kt + 1 + SWIFTK_X_TOUCH, nk_all - 1);
                toff = (uint32_t)min(tk, nk_all - 1) * ROWB;
            }
            auto k_half = [&](const int ch, const bool with_dma, const bool with_touch) {
                uint4 wf = *reinterpret_cast<const uint4*>(s + woff + ch);
#if SWIFTK_X_PF2
                uint4 wf1 = *reinterpret_cast<const uint4*>(s + woff + 16 * ROWB + ch);  // W fragments run two steps ahead
#endif
#pragma unroll
                for (int j = 0; j < NI; ++j) {
#if SWIFTK_X_PF2
                    uint4 wn_ = wf1;
#if SWIFTK_X_FEWREADS  // timing probe, WRONG results: only every FEWREADS-th W fragment is read, the others reuse the previous one
                    if (j + 2 < NI && (j + 2) % SWIFTK_X_FEWREADS == 0) wf1 = *reinterpret_cast<const uint4*>(s + woff + (j + 2) * 16 * ROWB + ch);
#else
                    if (j + 2 < NI) wf1 = *reinterpret_cast<const uint4*>(s + woff + (j + 2) * 16 * ROWB + ch);
#endif
#else
                    uint4 wn_ = wf;
                    if (j + 1 < NI) wn_ = *reinterpret_cast<const uint4*>(s + woff + (j + 1) * 16 * ROWB + ch);
#endif
#pragma unroll
                    for (int i = 0; i < MI; ++i) mma_chunk<T>(acc[i][j], wf, xf[i]);
#if SWIFTK_GEMM_INSTR
                    if (with_dma && j < 10 && !(g.dbg & 1)) issue_piece(fill, koff, j);
#else
                    if (with_dma && j < 10) issue_piece(fill, koff, j);
#endif
                    if constexpr (HPF) {
                        if (with_dma && last_k && j < 6) {  // line li of the wave's 32 rows x 11 lines (row-major), one per lane
                            int el = lane;
                            asm volatile("" : "+v"(el));
                            const int li = min(el + 64 * j, 351), r = li / 11, cl = li - 11 * r;
                            dma_touch(lds0 + 2 * STAGE, hpf_base, (uint32_t)(min(r, hpf_rows) * (int)g.pos_rows * 2 + min(cl * 128, hpf_cols)));
                        }
                    }
                    if constexpr (TOUCH) {
                        if (with_touch && j == 2) {
                            if (wv < 4) dma_touch(lds0 + 2 * STAGE, g.A + (int64_t)t_m0 * g.lda_b, (uint32_t)(min(tr1, g.M - 1 - t_m0) * (int)g.lda_b) + toff);
                            else dma_touch(lds0 + 2 * STAGE, g.W + (int64_t)t_n0 * g.ldw_b, (uint32_t)(min(tr1, g.N - 1 - t_n0) * (int)g.ldw_b) + toff);
                        }
                        if (with_touch && j == 6)
                            dma_touch(lds0 + 2 * STAGE, g.W + (int64_t)t_n0 * g.ldw_b, (uint32_t)(min(tr2, g.N - 1 - t_n0) * (int)g.ldw_b) + toff);
                    }
                    wf = wn_;
                }
            };
            k_half(ch0, true, false);
            if (!half) {
#pragma unroll
                for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const uint4*>(s + xoff + i * 16 * ROWB + ch1);
                k_half(ch1, false, TOUCH);
            }
        }
        par ^= 1;
        // Two-level accumulation (fp32 operands, g.kchunk > 0).  One MFMA chain over K = 1056 .. 2816 rounds 264 .. 704 times
        // at the growing partial sum's magnitude and ends 1.8 x further from the fp64 product than the CPU's blocked sgemm
        // (tests/fp32_bisect.py: the GEMMs are the ONLY op family of the fp32 engine less accurate than ATen's).  Chains of
        // `kchunk` k-tiles, summed through the workgroup's cache-resident scratch slab, bring that to the CPU's level; a
        // 256 x 352 fp32 tile takes ~300 us, parking 352 KB a few times per tile is not measurable.  Written like the ACCUM
        // epilogue (groups of four blocks, the next group's loads in flight) and placed behind the k-step's own wait, so the
        // k-loop's register allocation is the plain kernel's.
        auto chunk_io = [&](const bool rd, const bool wr) {
            int elane = lane;  // (opaque copy: hipcc otherwise hoists the 44 slab addresses above the k-loop and spills them)
            asm volatile("" : "+v"(elane));
            float4* scr = reinterpret_cast<float4*>(g.kscr) + ((int64_t)blockIdx.x * 8 + wv) * (MI * NI) * 64 + elane;
            constexpr int JH = 4, GPI = (NI + JH - 1) / JH, NG = MI * GPI;
            float4 buf[2][JH];
            auto load_group = [&](int gidx, float4 (&b)[JH]) {
                const int i = gidx / GPI, jh = (gidx - i * GPI) * JH;
#pragma unroll
                for (int jj = 0; jj < JH; ++jj) {
                    b[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (jh + jj < NI && rd) b[jj] = scr[(i * NI + jh + jj) * 64];
                }
            };
            load_group(0, buf[0]);
#pragma unroll
            for (int gidx = 0; gidx < NG; ++gidx) {
                if (gidx + 1 < NG) load_group(gidx + 1, buf[(gidx + 1) & 1]);
                const int i = gidx / GPI, jh = (gidx - i * GPI) * JH;
#pragma unroll
                for (int jj = 0; jj < JH; ++jj) {
                    if (jh + jj < NI) {
                        const int j = jh + jj;
                        const float4 o = buf[gidx & 1][jj];
                        f32x4 v = acc[i][j];
                        v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                        if (wr) {
                            scr[(i * NI + j) * 64] = make_float4(v[0], v[1], v[2], v[3]);
                            v = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        acc[i][j] = v;
                    }
                }
            }
        };
        bool chain_end = last_k;
        if constexpr (sizeof(T) == 4) chain_end = last_k || (g.kchunk > 0 && (kt + 1 - k_begin(tile)) % g.kchunk == 0);
        if (!chain_end) {
            ++kt;
            first_kt = false;
#if SWIFTK_PP_STAMP
            ++pl_kt;
#endif
            // (a k-tile that is not its tile's last is never the half one: both k-halves ran, so with the look-ahead on its two
            // requests are this wave's youngest VMEM operations and stay in flight)
            if constexpr (PP) {}  // (waited in the k-tile's last MEM phase)
            else if constexpr (TOUCH) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        if constexpr (PP) {
            // group 0 finished its MFMAs one barrier ahead of group 1: meet it, every wave is aligned again for the epilogue
            if (!grp) __builtin_amdgcn_s_barrier();
            first_kt = true;
#if SWIFTK_PP_STAMP
            ++pl_tile;
            pl_kt = 0;
#endif
        }
        if constexpr (sizeof(T) == 4) {
            // end of a chain that is not the tile's last: park the partial sum (added to what is parked already);
            // the tile's last chain: add what the earlier ones parked, then the epilogue proper.  ONE call site.
            if (have_part || !last_k) {
                chunk_io(have_part, !last_k);
                have_part = !last_k;
            }
            if (!last_k) {
                ++kt;
                // VMEM retires in issue order: the next stage's DMA pieces are older than the park's MI x NI stores, so leaving
                // exactly those outstanding is enough -- they drain under the next chain's MFMAs
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NI) : "memory");
                continue;
            }
        }
        bool interior = false;
        // ---- epilogue of `tile`: lane holds C[m][nb .. nb+3] for m = ..+r16, nb = ..+4*(lane>>4) ----
#if SWIFTK_GEMM_INSTR || SWIFTK_PP_STAMP
#if SWIFTK_GEMM_INSTR
        if (tlog) tlog[tl_i * 8 + 1] = __builtin_amdgcn_s_memtime();
        first_k = true;
#endif
        if (g.dbg & 4) {  // tuning experiment: drop the epilogue (keeps the accumulators live through a fake use)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    asm volatile("" ::"v"(acc[i][j]));
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        } else
#endif
        {
            int tm, tn;
            it.coords(tile_of(tile), tm, tn);
            const int m0 = tm * BM, n0 = tn * BN;
            if constexpr (EPI == SWIFTK_EPI_QKNORM || EPI == EPI_QKNORM_TILED)
                qknorm_tile<NI>(acc, lane, n0 + wn * WT, g.ep0, const_cast<float*>(g.ep1), m0 + wm * 64, g.M, g.N / HD, sizeof(T) == 4 && g.qk_only);
            float qf[MI / 2][4];  // QKNORM_JVP: (f_A, f_B, f c_A, f c_B) per primal row block
            if constexpr (EPI == SWIFTK_EPI_QKNORM_JVP) {
                int qlane = lane;  // (opaque copy: nothing of the tangent rule is hoisted above the k-loop)
                asm volatile("" : "+v"(qlane));
                qknorm_jvp_stats<NI>(acc, qlane, n0 + wn * WT, g.ep0, const_cast<float*>(g.ep1), tm * (BM / 2) + wm * 32, g.N / HD, qf);
            }
            OutT* C = reinterpret_cast<OutT*>(g.C) + (int64_t)split_of(tile) * g.c_split;
            if constexpr (EPI == SWIFTK_EPI_SWIGLU_JVP) {
                // FeedForward gate and its tangent (swinv2.py:99-100 under jvp): per 16-token group the primal pre-activations
                // (optional: the backward pass's saved activation), then silu(gate) * up and its tangent side by side in the
                // wave's slab, rows leaving as whole 16-B chunks like every bf16 tile
                constexpr int RS1 = WT * 2 + 16, RS2 = WT + 16;  // padded slab row strides (bytes): WT / WT/2 columns
                constexpr int CP1 = WT / 8, CP2 = WT / 16;       // 16-B chunks per row
                constexpr int SL = 16 * RS1 > 32 * RS2 ? 16 * RS1 : 32 * RS2;
                __builtin_amdgcn_s_barrier();
                char* slab = const_cast<char*>(s) + wv * SL;
                int elane = lane;
                asm volatile("" : "+v"(elane));
                const int g4 = elane >> 4;
                const int r16 = elane & 15;
                bf16_t* C2 = reinterpret_cast<bf16_t*>(const_cast<float*>(g.ep1));
                const int64_t ldc2 = g.pos_rows;
                const int mh = g.M >> 1;
#pragma unroll
                for (int i = 0; i < MI / 2; ++i) {
                    const int mrow0 = tm * (BM / 2) + wm * 32 + i * 16;
                    if (g.C) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const f32x4 v = acc[i][j];
                            *reinterpret_cast<uint2*>(slab + r16 * RS1 + (j * 16 + 4 * g4) * 2) =
                                make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int t = 0; t < (16 * CP1 + 63) / 64; ++t) {
                            const int c = elane + 64 * t;
                            const int row = c / CP1, cc = c - row * CP1;
                            if (c < 16 * CP1) {
                                const uint4 q = *reinterpret_cast<const uint4*>(slab + row * RS1 + cc * 16);
                                const int n = n0 + wn * WT + cc * 8;
                                if (n < g.N) *reinterpret_cast<uint4*>(C + (int64_t)(mrow0 + row) * g.ldc + n) = q;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const f32x4 v = acc[i][j], t = acc[i + 2][j];
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[i + 2][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const float s0 = __builtin_amdgcn_rcpf(1.0f + __expf(-v[0])), s1 = __builtin_amdgcn_rcpf(1.0f + __expf(-v[2]));
                        const float a0 = v[0] * s0, a1 = v[2] * s1;  // silu(gate)
                        *reinterpret_cast<uint32_t*>(slab + r16 * RS2 + (j * 8 + 2 * g4) * 2) = pack_bf16(a0 * v[1], a1 * v[3]);
                        *reinterpret_cast<uint32_t*>(slab + (16 + r16) * RS2 + (j * 8 + 2 * g4) * 2) =
                            pack_bf16((s0 + a0 * (1.0f - s0)) * t[0] * v[1] + a0 * t[1], (s1 + a1 * (1.0f - s1)) * t[2] * v[3] + a1 * t[3]);
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < (32 * CP2 + 63) / 64; ++t) {
                        const int c = elane + 64 * t;
                        const int row = c / CP2, cc = c - row * CP2;  // rows 0..15: silu(gate) up, 16..31: its tangent
                        if (c < 32 * CP2) {
                            const uint4 q = *reinterpret_cast<const uint4*>(slab + row * RS2 + cc * 16);
                            const int m = mrow0 + (row & 15) + (row >= 16 ? mh : 0), n = (n0 >> 1) + wn * (WT / 2) + cc * 8;
                            if (n < (g.N >> 1)) *reinterpret_cast<uint4*>(C2 + (int64_t)m * ldc2 + n) = q;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else
            if constexpr (EPI == SWIFTK_EPI_SWIGLU_BWD) {
                // backward of the FeedForward's gate: the accumulators are d(hidden)[m][j]; with the saved pre-activation
                // (gate, up) = H[m][2j], H[m][2j+1] the tile leaves as d(pre-activation)[m][2j .. 2j+1] -- a lane's four
                // columns are 16 contiguous bytes of H and of the output, so no LDS pass; d(hidden) never reaches memory
                const bf16_t* H = reinterpret_cast<const bf16_t*>(g.ep1);
                const int64_t ldh = g.pos_rows;
                int elane = lane;
                asm volatile("" : "+v"(elane));
                const int g4 = elane >> 4;
                // groups of four 16-column blocks, the saved pre-activations of group g+1 requested before group g is computed
                constexpr int JH = 4, GPI = (NI + JH - 1) / JH, NG = MI * GPI;
                uint4 hb[2][JH];
                auto load_group = [&](int gidx, uint4 (&b)[JH]) {
                    const int i = gidx / GPI, jh = (gidx - i * GPI) * JH;
                    const int m = m0 + wm * 64 + i * 16 + (elane & 15);
#pragma unroll
                    for (int jj = 0; jj < JH; ++jj) {
                        const int nb = n0 + wn * WT + (jh + jj) * 16 + 4 * g4;
                        b[jj] = make_uint4(0u, 0u, 0u, 0u);
                        if (jh + jj < NI && m < g.M && nb < g.N) b[jj] = *reinterpret_cast<const uint4*>(H + (int64_t)m * ldh + 2 * nb);
                    }
                };
#if SWIFTK_X_BWDROLL
                // (round 6 experiment) a ROLLING window over the tile's MI x NI saved pre-activation chunks: chunk q + R is requested the
                // moment chunk q has been consumed, so R loads per lane stay in flight throughout (the grouped form swings between 4 and 8)
                constexpr int R = SWIFTK_X_BWDROLL;
                uint4 hr[R];
                auto load_one = [&](int q) -> uint4 {
                    const int i = q / NI, j = q - i * NI;
                    const int m = m0 + wm * 64 + i * 16 + (elane & 15);
                    const int nb = n0 + wn * WT + j * 16 + 4 * g4;
                    uint4 r = make_uint4(0u, 0u, 0u, 0u);
                    if (m < g.M && nb < g.N) r = *reinterpret_cast<const uint4*>(H + (int64_t)m * ldh + 2 * nb);
                    return r;
                };
#pragma unroll
                for (int q = 0; q < R; ++q) hr[q] = load_one(q);
#pragma unroll
                for (int q = 0; q < MI * NI; ++q) {
                    const int i = q / NI, j = q - i * NI;
                    const int m = m0 + wm * 64 + i * 16 + (elane & 15);
                    const int nb = n0 + wn * WT + j * 16 + 4 * g4;
                    const f32x4 v = acc[i][j];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const uint4 hq = hr[q % R];
                    if (q + R < MI * NI) hr[q % R] = load_one(q + R);
                    const uint32_t hw[4] = {hq.x, hq.y, hq.z, hq.w};
                    uint32_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gt = __uint_as_float(hw[e] << 16), up = __uint_as_float(hw[e] & 0xffff0000u);
                        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-gt));
                        o[e] = pack_bf16(v[e] * up * (sg + gt * sg * (1.0f - sg)), v[e] * gt * sg);
                    }
                    if (m < g.M && nb < g.N)
                        *reinterpret_cast<uint4*>(C + (int64_t)m * g.ldc + 2 * nb) = make_uint4(o[0], o[1], o[2], o[3]);
                }
#else
                load_group(0, hb[0]);
#pragma unroll
                for (int gidx = 0; gidx < NG; ++gidx) {
                    if (gidx + 1 < NG) load_group(gidx + 1, hb[(gidx + 1) & 1]);
                    const int i = gidx / GPI, jh = (gidx - i * GPI) * JH;
                    const int m = m0 + wm * 64 + i * 16 + (elane & 15);
#pragma unroll
                    for (int jj = 0; jj < JH; ++jj) {
                        if (jh + jj < NI) {
                            const int j = jh + jj;
                            const int nb = n0 + wn * WT + j * 16 + 4 * g4;
                            const f32x4 v = acc[i][j];
                            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                            const uint4 hq = hb[gidx & 1][jj];
                            const uint32_t hw[4] = {hq.x, hq.y, hq.z, hq.w};
                            uint32_t o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float gt = __uint_as_float(hw[e] << 16), up = __uint_as_float(hw[e] & 0xffff0000u);
                                const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-gt));
                                o[e] = pack_bf16(v[e] * up * (sg + gt * sg * (1.0f - sg)), v[e] * gt * sg);
                            }
                            if (m < g.M && nb < g.N)
                                *reinterpret_cast<uint4*>(C + (int64_t)m * g.ldc + 2 * nb) = make_uint4(o[0], o[1], o[2], o[3]);
                        }
                    }
                }
#endif
            } else
            if constexpr (EPI == SWIFTK_EPI_SWIGLU_BOTH) {
                // training forward: the pre-activation h (the backward pass needs gate and up) AND silu(gate) * up leave in
                // one epilogue -- two passes through the wave's LDS slab per 16-row group, 1.5x the stores of a plain tile,
                // instead of a second kernel that re-reads h (3.4 % of a CRPS iteration)
                constexpr int RS1 = WT * 2 + 16, RS2 = WT + 16;  // padded slab row strides (bytes): WT / WT/2 columns
                constexpr int CP1 = WT / 8, CP2 = WT / 16;       // 16-B chunks per row
                __builtin_amdgcn_s_barrier();
#if SWIFTK_X_VMCNT
                interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
#endif
                char* slab = const_cast<char*>(s) + wv * (16 * RS1);
                int elane = lane;
                asm volatile("" : "+v"(elane));
                const int g4 = elane >> 4;
                const int r16 = elane & 15;
                bf16_t* C2 = reinterpret_cast<bf16_t*>(const_cast<float*>(g.ep1));
                const int64_t ldc2 = g.pos_rows;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int mrow0 = m0 + wm * 64 + i * 16;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const f32x4 v = acc[i][j];
                        *reinterpret_cast<uint2*>(slab + r16 * RS1 + (j * 16 + 4 * g4) * 2) =
                            make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < (16 * CP1 + 63) / 64; ++t) {
                        const int c = elane + 64 * t;
                        const int row = c / CP1, cc = c - row * CP1;
                        if (c < 16 * CP1) {
                            const uint4 q = *reinterpret_cast<const uint4*>(slab + row * RS1 + cc * 16);
                            const int m = mrow0 + row, n = n0 + wn * WT + cc * 8;
                            if (m < g.M && n < g.N) *reinterpret_cast<uint4*>(C + (int64_t)m * g.ldc + n) = q;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const f32x4 v = acc[i][j];
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const float h0 = v[0] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[0])) * v[1];
                        const float h1 = v[2] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[2])) * v[3];
                        *reinterpret_cast<uint32_t*>(slab + r16 * RS2 + (j * 8 + 2 * g4) * 2) = pack_bf16(h0, h1);
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < (16 * CP2 + 63) / 64; ++t) {
                        const int c = elane + 64 * t;
                        const int row = c / CP2, cc = c - row * CP2;
                        if (c < 16 * CP2) {
                            const uint4 q = *reinterpret_cast<const uint4*>(slab + row * RS2 + cc * 16);
                            const int m = mrow0 + row, n = (n0 >> 1) + wn * (WT / 2) + cc * 8;
                            if (m < g.M && n < (g.N >> 1)) *reinterpret_cast<uint4*>(C2 + (int64_t)m * ldc2 + n) = q;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else
            if constexpr (sizeof(OutT) == 2 && EPI != SWIFTK_EPI_BIAS_POS) {
                // bf16 output: transpose each 16-row slab of the wave's tile through LDS so that rows leave as whole
                // 16-B chunks (one dwordx4 store covers 5.8 contiguous rows' worth) instead of 44 scattered 4..8-B
                // stores per wave -- the scattered form is store-issue bound (4.6 us per tile, 14.5 us with SwiGLU).
                // The stage just consumed (`s`) is free: its refill is issued only after the next barrier, which no
                // wave passes before every wave has finished this epilogue.  Slabs are wave-private: no barrier.
                constexpr bool SPLIT3 = EPI == SWIFTK_EPI_SWIGLU_SPLIT3;
                constexpr bool PAIROUT = EPI == EPI_BIAS_POS_PAIR;
                constexpr bool GLU = EPI == SWIFTK_EPI_SWIGLU || SPLIT3;
                constexpr int COLS = GLU ? WT / 2 : WT;  // output columns of the wave tile
                constexpr int CPR = COLS / 8;                                  // 16-B chunks per row
                constexpr int RSTR = COLS * 2 + 16;                            // padded slab row stride (bytes)
                // the slabs overlay operand bytes of the stage just consumed: every wave must be done READING that stage
                // (its last fragments were consumed by MFMAs it has already issued) before any wave writes a slab
                __builtin_amdgcn_s_barrier();
#if SWIFTK_GEMM_INSTR
                if (tlog) tlog[tl_i * 8 + 2] = __builtin_amdgcn_s_memtime();
#endif
#if SWIFTK_X_VMCNT
                // interior tile: every lane group of every store below is live, so the wave issues exactly 24 (12 with
                // SwiGLU) global stores after its last DMA piece -- the count the next loop trip leaves outstanding
                interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
#endif
                char* slab = const_cast<char*>(s) + wv * (16 * RSTR);
                // lane-derived epilogue addresses are rebuilt from an opaque copy of the lane id: left visible, hipcc
                // hoists ~40 loop-invariant epilogue VGPRs above the k-loop and spills them inside it
                int elane = lane;
                asm volatile("" : "+v"(elane));
                const int g4 = elane >> 4;
                const int r16 = elane & 15;
                const int ncol0 = (GLU ? (n0 >> 1) : n0) + wn * COLS;
                const int nout = GLU ? (g.N >> 1) : g.N;
                uint32_t lo_pk[SPLIT3 || PAIROUT ? NI : 1];  // SPLIT3 / PAIROUT: the low parts of this row block's (hi, lo) pairs
#pragma unroll
                for (int ii = 0; ii < MI; ++ii) {
                    // QKNORM_JVP: a tangent row block leaves before its primal block (2, 0, 3, 1) -- its rule reads the primal values
                    const int i = EPI == SWIFTK_EPI_QKNORM_JVP ? ((ii & 1) ? ii >> 1 : 2 + (ii >> 1)) : ii;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        f32x4 v = acc[i][j];
                        if constexpr (EPI == SWIFTK_EPI_QKNORM_JVP) {
                            const bool first = 16 * j + 12 < 8 * NI ? true : (16 * j >= 8 * NI ? false : 16 * j + 4 * g4 < 8 * NI);
                            const float f = first ? qf[i & 1][0] : qf[i & 1][1];
                            if (i >= 2) {
                                const float fc = first ? qf[i & 1][2] : qf[i & 1][3];
                                const f32x4 pv = acc[i & 1][j];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaf(-fc, pv[e], f * v[e]);
                            } else {
                                v *= f;
                            }
                        }
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (PAIROUT) {
                            // patch embedding straight into the pair form of the residual stream: x = acc + bias + pos in fp32 (the
                            // fp32-output epilogue's order), hi = bf16(x) through the slab, the 8-bit low parts follow in a second pass
                            const int nb = n0 + wn * WT + j * 16 + 4 * g4;
                            const int m = m0 + wm * 64 + i * 16 + r16;
                            if (m < g.M && nb < g.N) {
                                const float4 b = *reinterpret_cast<const float4*>(g.ep0 + nb);
                                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                                if (g.ep1) {
                                    const float4 pp = *reinterpret_cast<const float4*>(g.ep1 + (int64_t)(m % g.pos_rows) * g.N + nb);
                                    v[0] += pp.x; v[1] += pp.y; v[2] += pp.z; v[3] += pp.w;
                                }
                            }
                            const uint32_t h0 = pack_bf16(v[0], v[1]), h1 = pack_bf16(v[2], v[3]);
                            uint32_t lb = 0u;
                            lb = lo8_insert(v[0], __uint_as_float(h0 << 16), (h0 >> 7) & 0xFFu, 0, lb);
                            lb = lo8_insert(v[1], __uint_as_float(h0 & 0xffff0000u), (h0 >> 23) & 0xFFu, 1, lb);
                            lb = lo8_insert(v[2], __uint_as_float(h1 << 16), (h1 >> 7) & 0xFFu, 2, lb);
                            lb = lo8_insert(v[3], __uint_as_float(h1 & 0xffff0000u), (h1 >> 23) & 0xFFu, 3, lb);
                            lo_pk[j] = lb;
                            *reinterpret_cast<uint2*>(slab + r16 * RSTR + (j * 16 + 4 * g4) * 2) = make_uint2(h0, h1);
                        } else if constexpr (SPLIT3) {
                            // the split engine's hidden activation leaves as the NEXT GEMM's operand blocks: hi = bf16(h),
                            // lo = bf16(h - hi) (fp32-grade silu as in the fp32-output form; h itself never reaches memory)
                            const float h0 = swiglu_out<float>(v[0], v[1]), h1 = swiglu_out<float>(v[2], v[3]);
                            const uint32_t hp = pack_bf16(h0, h1);
                            lo_pk[j] = pack_bf16(h0 - __uint_as_float(hp << 16), h1 - __uint_as_float(hp & 0xffff0000u));
                            *reinterpret_cast<uint32_t*>(slab + r16 * RSTR + (j * 8 + 2 * g4) * 2) = hp;
                        } else if constexpr (EPI == SWIFTK_EPI_SWIGLU) {
                            // silu(g) * u with v_exp_f32 / v_rcp_f32 (1 ulp each; the libm forms cost ~30 VALU apiece)
#if SWIFTK_X_NOSILU  // timing probe with WRONG results (round 6): the epilogue without its transcendentals = the bound of any cheaper sigmoid
                            const float h0 = v[0] * v[1];
                            const float h1 = v[2] * v[3];
#else
                            const float h0 = v[0] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[0])) * v[1];
                            const float h1 = v[2] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[2])) * v[3];
#endif
                            *reinterpret_cast<uint32_t*>(slab + r16 * RSTR + (j * 8 + 2 * g4) * 2) = pack_bf16(h0, h1);
                        } else {
                            *reinterpret_cast<uint2*>(slab + r16 * RSTR + (j * 16 + 4 * g4) * 2) =
                                make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();  // slab written (LDS ops of a wave execute in order)
                    const int mrow0 = PAIRED ? (i >= 2 ? (g.M >> 1) : 0) + tm * (BM / 2) + wm * 32 + (i & 1) * 16 : m0 + wm * 64 + i * 16;
                    // Window-tiled destination (to_qkv for the streamed attention kernel): the 16 rows of a slab are 16
                    // consecutive tokens of one grid row (gw % 16 == 0), so sample / grid row / column base are
                    // wave-uniform per slab; a row's 88-wide q, k or v slice lands at [window][head][part][idx][0..87],
                    // where the window partition is that of the grid rolled by (-t_sh, -t_sw) (swinv2.py:185-189).
                    int tb = 0, try_ = 0, tx16 = 0, twin0 = 0;
                    if constexpr (EPI == EPI_QKNORM_TILED) {
                        {
                            const int ntok = g.t_gh * g.t_gw;
                            tb = mrow0 / ntok;
                            const int tt = mrow0 - tb * ntok, y = tt / g.t_gw;
                            tx16 = tt - y * g.t_gw;
                            try_ = y - g.t_sh;
                            try_ += try_ < 0 ? g.t_gh : 0;
                            twin0 = tb * ((g.t_gh >> 4) * (g.t_gw >> 4)) + (try_ >> 4) * (g.t_gw >> 4);
                        }
                    }
                    constexpr int NCH = (16 * CPR + 63) / 64;
#if SWIFTK_X_EPIBATCH
                    uint4 qs[NCH];
#pragma unroll
                    for (int t = 0; t < NCH; ++t) {
                        const int c = min(elane + 64 * t, 16 * CPR - 1);
                        const int row = c / CPR, cc = c - row * CPR;
                        qs[t] = *reinterpret_cast<const uint4*>(slab + row * RSTR + cc * 16);
                    }
#endif
#pragma unroll
                    for (int t = 0; t < NCH; ++t) {
                        const int c = elane + 64 * t;
                        const int row = c / CPR, cc = c - row * CPR;
                        if (c < 16 * CPR) {
#if SWIFTK_X_EPIBATCH
                            const uint4 q = qs[t];
#else
                            const uint4 q = *reinterpret_cast<const uint4*>(slab + row * RSTR + cc * 16);
#endif
                            const int m = mrow0 + row, n = ncol0 + cc * 8;
                            int64_t dst = (int64_t)m * g.ldc + n;
                            if constexpr (EPI == EPI_QKNORM_TILED) {
                                {
                                    int rx = tx16 + row - g.t_sw;
                                    rx += rx < 0 ? g.t_gw : 0;
                                    const int hi = cc >= NI;  // NI 16-B chunks per head vector
                                    // tile = ((sample * windows + window) * heads + head) * 3 + part; the last two terms
                                    // are the row's 88-wide slice number 4 * tn + 2 * wn + hi
                                    const int tile_ = (twin0 + (rx >> 4)) * (3 * g.t_heads) + tn * 4 + wn * 2 + hi;
                                    dst = (int64_t)tile_ * (256 * HD) + ((((try_ & 15) << 4) | (rx & 15)) * HD + (cc - NI * hi) * 8);
                                }
                            }
                            if (m < g.M && n < nout) store16_out(C + dst, q);
                            if constexpr (SPLIT3) {  // [hi | lo | hi]: the hi block twice (blocks g.pos_rows columns apart)
                                if (m < g.M && n < nout) store16_out(C + dst + 2 * (int64_t)g.pos_rows, q);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();  // slab read before the next slab overwrites it
                    if constexpr (SPLIT3) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) *reinterpret_cast<uint32_t*>(slab + r16 * RSTR + (j * 8 + 2 * g4) * 2) = lo_pk[j];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int t = 0; t < (16 * CPR + 63) / 64; ++t) {
                            const int c = elane + 64 * t;
                            const int row = c / CPR, cc = c - row * CPR;
                            if (c < 16 * CPR) {
                                const uint4 q = *reinterpret_cast<const uint4*>(slab + row * RSTR + cc * 16);
                                const int m = mrow0 + row, n = ncol0 + cc * 8;
                                if (m < g.M && n < nout) store16_out(C + (int64_t)m * g.ldc + n + g.pos_rows, q);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    if constexpr (PAIROUT) {
                        constexpr int RB = WT + 16, CB = WT / 16;  // byte rows of the low parts: WT bytes = CB 16-B chunks
                        uint8_t* lo = reinterpret_cast<uint8_t*>(g.kscr);
#pragma unroll
                        for (int j = 0; j < NI; ++j) *reinterpret_cast<uint32_t*>(slab + r16 * RB + j * 16 + 4 * g4) = lo_pk[j];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int t = 0; t < (16 * CB + 63) / 64; ++t) {
                            const int c = elane + 64 * t;
                            const int row = c / CB, cc = c - row * CB;
                            if (c < 16 * CB) {
                                const uint4 q = *reinterpret_cast<const uint4*>(slab + row * RB + cc * 16);
                                const int m = mrow0 + row, n = n0 + wn * WT + cc * 16;
                                if (m < g.M && n < g.N) store16_out(lo + (int64_t)m * g.c_split + n, q);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            } else if constexpr (EPI == SWIFTK_EPI_ACCUM) {
                // C += A W^T (fp32): the residual-stream gradient picks up a branch's input gradient in the GEMM that
                // produces it.  The tile is walked in groups of four 16-column blocks; the loads of group g+1 are issued
                // before group g is added and stored (same array: the compiler keeps load g+1 behind the stores of g-1,
                // so one group of look-ahead is what program order allows) -- one exposed load latency per tile, not twelve.
                constexpr int JH = 4, GPI = (NI + JH - 1) / JH, NG = MI * GPI;
                float4 buf[2][JH];
                auto load_group = [&](int gidx, float4 (&b)[JH]) {
                    const int i = gidx / GPI, jh = (gidx - i * GPI) * JH;
                    const int m = m0 + wm * 64 + i * 16 + r16;
#pragma unroll
                    for (int jj = 0; jj < JH; ++jj) {
                        const int nb = n0 + wn * WT + (jh + jj) * 16 + 4 * (lane >> 4);
                        b[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (jh + jj < NI && m < g.M && nb < g.N) b[jj] = *reinterpret_cast<const float4*>(C + (int64_t)m * g.ldc + nb);
                    }
                };
                load_group(0, buf[0]);
#pragma unroll
                for (int gidx = 0; gidx < NG; ++gidx) {
                    if (gidx + 1 < NG) load_group(gidx + 1, buf[(gidx + 1) & 1]);
                    const int i = gidx / GPI, jh = (gidx - i * GPI) * JH;
                    const int m = m0 + wm * 64 + i * 16 + r16;
#pragma unroll
                    for (int jj = 0; jj < JH; ++jj) {
                        if (jh + jj < NI) {
                            const int j = jh + jj;
                            const int nb = n0 + wn * WT + j * 16 + 4 * (lane >> 4);
                            const f32x4 v = acc[i][j];
                            const float4 o = buf[gidx & 1][jj];
                            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (m < g.M && nb < g.N) store4<OutT>(C + (int64_t)m * g.ldc + nb, v[0] + o.x, v[1] + o.y, v[2] + o.z, v[3] + o.w);
                        }
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * 64 + i * 16 + r16;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int nb = n0 + wn * WT + j * 16 + 4 * (lane >> 4);
                    f32x4 v = acc[i][j];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (m >= g.M || nb >= g.N) continue;
                    if constexpr (EPI == SWIFTK_EPI_BIAS_POS) {
                        const float4 b = *reinterpret_cast<const float4*>(g.ep0 + nb);
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                        if (g.ep1) {
                            const float4 pp =
                                *reinterpret_cast<const float4*>(g.ep1 + (int64_t)(m % g.pos_rows) * g.N + nb);
                            v[0] += pp.x; v[1] += pp.y; v[2] += pp.z; v[3] += pp.w;
                        }
                    }
                    if constexpr (EPI == SWIFTK_EPI_SWIGLU) {
                        const float h0 = swiglu_out<OutT>(v[0], v[1]);
                        const float h1 = swiglu_out<OutT>(v[2], v[3]);
                        store2<OutT>(C + (int64_t)m * g.ldc + (nb >> 1), h0, h1);
                    } else {
                        int nbo = nb;
                        if constexpr (sizeof(T) == 4 && EPI == SWIFTK_EPI_QKNORM)  // ... and land in the q, k columns of the [q | k | v] output
                            if (g.qk_only) nbo += (nb / (2 * HD)) * HD;
                        store4<OutT>(C + (int64_t)m * g.ldc + nbo, v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
#if SWIFTK_GEMM_INSTR
        if (tlog) {
            tlog[tl_i * 8 + 3] = __builtin_amdgcn_s_memtime();
            ++tl_i;
        }
#endif
        tile += stride;
        if (tile >= ntiles) break;
        kt = k_begin(tile);
        nk = k_end(tile);
        // VMEM retires in issue order: the DMA pieces of the next tile's first stage are older than this tile's epilogue
        // stores, so leaving exactly those stores outstanding is enough (no store drain in front of a tile)
        if (interior) {
            // stores per wave of an interior tile: 4 slabs x ceil(16 rows x (COLS / 8) chunks / 64 lanes)
            constexpr int NSTORE = (EPI == EPI_BIAS_POS_PAIR ? 4 * ((16 * (WT / 16) + 63) / 64) : 0) + (EPI == SWIFTK_EPI_SWIGLU_SPLIT3 ? 12 : 4) *
                                       ((16 * ((EPI == SWIFTK_EPI_SWIGLU || EPI == SWIFTK_EPI_SWIGLU_SPLIT3 ? WT / 2 : WT) / 8) + 63) / 64) +
                                   (EPI == SWIFTK_EPI_SWIGLU_BOTH ? 4 * ((16 * (WT / 16) + 63) / 64) : 0);
            if constexpr (sizeof(OutT) == 2 && EPI != SWIFTK_EPI_BIAS_POS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing dummy DMA must not outlive the LDS allocation
}

// ---- optional live timing of one GEMM flavour (bench.py's roofline leg): HIP events on the launch stream ----
int g_variant = 1;   // 0: one tile per workgroup; 1: persistent, grouped tile order, interleaved DMA
int g_group_m = 8;   // tile rows per group in the persistent order
int g_dbg = 0;
int g_stagger_permille = 0;  // tuning key 7: start-up phase step as a fraction (in 1/1000) of an eighth of the estimated tile time
int g_pp = SWIFTK_X_PP;      // tuning key 20: ping-pong k-loop of the persistent kernel (bf16 operands)

// the ping-pong loop prefetches W0 two k-tiles ahead: every work item needs at least three k-tiles
inline bool pp_ok(const GemmArgs& g) { return g_pp > 0 && !SWIFTK_GEMM_INSTR && (g.K / 64) / g.ksplit >= 3; }

struct Prof {
    int epilogue = -1, N = 0;
    static constexpr int MAXE = 4096;
    hipEvent_t ev[2 * MAXE];
    int created = 0, used = 0;
} g_prof;

template <typename T, typename OutT, int EPI>
int launch(const GemmArgs& g, hipStream_t st) {
    auto kern = gemm_kernel<T, OutT, EPI>;
    const int ntm = (g.M + BM - 1) / BM;
    if (sizeof(T) == 4 && EPI != SWIFTK_EPI_QKNORM && g.ni != NI) return SWIFTK_ESHAPE;  // (fp32 operands: 352-wide tiles but for QKNORM)
    // [q | k]-only form: fp32 operands and output, whole tiles, the persistent kernel (the only one that remaps W rows / C columns)
    if (g.qk_only && (sizeof(T) != 4 || sizeof(OutT) != 4 || EPI != SWIFTK_EPI_QKNORM || g_variant == 0 || (g.M & 7) || (g.N & 7) || g.ep1))
        return SWIFTK_ESHAPE;
    const bool timed = swiftk_prof_begin(EPI, g.N, st);
    const bool wide_ok = sizeof(OutT) != 2 || (!((uintptr_t)g.C & 15) && !(g.ldc & 7) &&
                                               !(g.N & (EPI == SWIFTK_EPI_SWIGLU || EPI == SWIFTK_EPI_SWIGLU_BOTH ? 15 : 7)));  // 16-B row chunks
    if (g_variant == 0 || (g.M & 7) || (g.N & 7) || !wide_ok) {  // ragged edges: per-lane clamped sources
        if (g.ksplit != 1 || g.t_gw || EPI == SWIFTK_EPI_ACCUM || EPI == SWIFTK_EPI_SWIGLU_BOTH || EPI == SWIFTK_EPI_SWIGLU_BWD)
            return SWIFTK_ESHAPE;
        if ((EPI == SWIFTK_EPI_QKNORM || EPI == EPI_QKNORM_TILED) && g.ni != NI) return SWIFTK_ESHAPE;  // 352-wide tiles only
        GemmArgs g1 = g;
        g1.ntn = (g.N + BN - 1) / BN;
        hipLaunchKernelGGL(kern, dim3(ntm * g1.ntn), dim3(NT), 0, st, g1);
    } else {
        const int ntiles = ntm * g.ntn * g.ksplit;
        const int grid = ntiles < g_persist_wgs ? ntiles : g_persist_wgs;
        if constexpr (sizeof(T) == 2) {  // bf16 operands: all three tile widths (head_dim 80 / 88 / 96 families)
            if (pp_ok(g)) {
                if (g.ni == 10) hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 10, true>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
                else if (g.ni == 12) hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 12, true>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
                else hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 11, true>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
            } else {
                if (g.ni == 10) hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 10, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
                else if (g.ni == 12) hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 12, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
                else hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 11, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
            }
        } else if constexpr (EPI == SWIFTK_EPI_QKNORM) {
            // fp32 operands: 352-wide tiles, and for the cosine-attention epilogue also the 320- / 384-wide ones (a tile must hold
            // whole head pairs: head_dim 80 / 96, the exact engine and the split engine's hot head pairs on the larger variants)
            if (g.ni == 10) hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 10, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
            else if (g.ni == 12) hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 12, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
            else hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 11, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
        } else {
            hipLaunchKernelGGL((gemm_kernel_p<T, OutT, EPI, 11, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
        }
    }
    if (timed) swiftk_prof_end(st);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

// the paired-row epilogues exist in the persistent kernel only (bf16 in, bf16 out, whole tiles)
template <int EPI>
int launch_paired(const GemmArgs& g, hipStream_t st) {
    const int ntm = (g.M + BM - 1) / BM;
    // (the pair-output epilogue keeps NI packed low parts beside the accumulators: at NI = 12 that is 204 of the 256 registers before
    // any address, and the 384-wide instantiation spilled 8-10 of them -- its caller takes 352-wide tiles instead)
    constexpr bool W384 = EPI != EPI_BIAS_POS_PAIR;
    if (!W384 && g.ni == 12) return SWIFTK_ESHAPE;
    const bool timed = swiftk_prof_begin(EPI, g.N, st);
    const int ntiles = ntm * g.ntn;
    const int grid = ntiles < g_persist_wgs ? ntiles : g_persist_wgs;
    if (pp_ok(g)) {
        if (g.ni == 10) hipLaunchKernelGGL((gemm_kernel_p<bf16_t, bf16_t, EPI, 10, true>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
        else if (W384 && g.ni == 12) hipLaunchKernelGGL((gemm_kernel_p<bf16_t, bf16_t, EPI, W384 ? 12 : 11, true>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
        else hipLaunchKernelGGL((gemm_kernel_p<bf16_t, bf16_t, EPI, 11, true>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
    } else {
        if (g.ni == 10) hipLaunchKernelGGL((gemm_kernel_p<bf16_t, bf16_t, EPI, 10, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
        else if (W384 && g.ni == 12) hipLaunchKernelGGL((gemm_kernel_p<bf16_t, bf16_t, EPI, W384 ? 12 : 11, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
        else hipLaunchKernelGGL((gemm_kernel_p<bf16_t, bf16_t, EPI, 11, false>), dim3(grid), dim3(NT), 0, st, g, ntm, g_group_m);
    }
    if (timed) swiftk_prof_end(st);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

template <typename T, typename OutT>
int dispatch_epi(int epi, const GemmArgs& g, hipStream_t st) {
    switch (epi) {
        case SWIFTK_EPI_NONE: return launch<T, OutT, SWIFTK_EPI_NONE>(g, st);
        case SWIFTK_EPI_BIAS_POS: return launch<T, OutT, SWIFTK_EPI_BIAS_POS>(g, st);
        case SWIFTK_EPI_SWIGLU: return launch<T, OutT, SWIFTK_EPI_SWIGLU>(g, st);
        case SWIFTK_EPI_SWIGLU_BWD:
            if constexpr (sizeof(OutT) == 2 && sizeof(T) == 2) return launch<T, OutT, SWIFTK_EPI_SWIGLU_BWD>(g, st);
            return SWIFTK_EINVAL;
        case SWIFTK_EPI_SWIGLU_BOTH:
            if constexpr (sizeof(OutT) == 2 && sizeof(T) == 2) return launch<T, OutT, SWIFTK_EPI_SWIGLU_BOTH>(g, st);
            return SWIFTK_EINVAL;
        case SWIFTK_EPI_ACCUM:
            if constexpr (sizeof(OutT) == 4) return launch<T, OutT, SWIFTK_EPI_ACCUM>(g, st);
            return SWIFTK_EINVAL;
        case SWIFTK_EPI_QKNORM:
            if constexpr (sizeof(OutT) == 2)
                if (g.t_gw) return launch<T, OutT, EPI_QKNORM_TILED>(g, st);
            return launch<T, OutT, SWIFTK_EPI_QKNORM>(g, st);
        case SWIFTK_EPI_SWIGLU_SPLIT3:
            if constexpr (sizeof(OutT) == 2 && sizeof(T) == 2) {
                if ((g.M & 7) || (g.N & 15) || ((uintptr_t)g.C & 15) || (g.ldc & 7) || (g.pos_rows & 7)) return SWIFTK_ESHAPE;
                return launch_paired<SWIFTK_EPI_SWIGLU_SPLIT3>(g, st);  // (persistent kernel only, like the paired-row epilogues)
            }
            return SWIFTK_EINVAL;
    }
    return SWIFTK_EINVAL;
}

}  // namespace

bool swiftk_prof_begin(int kind, int n, hipStream_t st) {
    if (g_prof.epilogue != kind || (g_prof.N != 0 && g_prof.N != n) || g_prof.used >= Prof::MAXE) return false;
    while (g_prof.created <= g_prof.used) {
        if (hipEventCreate(&g_prof.ev[2 * g_prof.created]) != hipSuccess) return false;
        if (hipEventCreate(&g_prof.ev[2 * g_prof.created + 1]) != hipSuccess) return false;
        ++g_prof.created;
    }
    return hipEventRecord(g_prof.ev[2 * g_prof.used], st) == hipSuccess;
}

void swiftk_prof_end(hipStream_t st) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], st);
    ++g_prof.used;
}

extern "C" int swiftk_set_tuning(int key, int value) {
    switch (key) {
        case 0: g_variant = value; return 0;
        case 1: g_group_m = value > 0 ? value : 1; return 0;
        case 2: g_persist_wgs = value > 0 ? value : 1; return 0;
        case 3: g_dbg = value; return 0;
        case 4: g_attn_dbg = value; return 0;
        case 5: g_fwd_tiled = value; return 0;
        case 6: g_modnorm_nt = value; return 0;
        case 7: g_stagger_permille = value; return 0;
        case 8: g_fwd_fused = value; return 0;
        case 9: g_attn_bwd_pipe = value; return 0;
        case 11: g_x3_exact = value; return 0;
        case 12: g_fwd_pair = value; return 0;
        case 13: g_f32_chunk_k = value; return 0;
        case 14: g_fwd_splitk = value; return 0;
        case 15: g_attn_bwd_fuse = value; return 0;
        case 16: g_modnorm_bwd_fused = value; return 0;
        case 17: g_modnorm_jvp_rows = value; return 0;
        case 18: g_x3_ffsplit = value; return 0;
        case 19: g_fwd_pepair = value; return 0;
        case 20: g_pp = value; return 0;
        case 21: g_attn_pp = value; return 0;
        case 22: g_tn_pp = value; return 0;
        case 23: g_fwd_rownorm = value; return 0;
        case 24: g_rownorm_dbg = value; return 0;
        case 26: g_x3_normsplit = value; return 0;
        case 27: g_x3_qkonly = value; return 0;
        case 28: g_x3_attnpv = value; return 0;
        case 29: g_fwd_tail = value; return 0;
        case 25:
            g_zero_memset = value;
            return (value & 4) ? swiftk_zero_check_enable() : 0;
    }
    return SWIFTK_EINVAL;
}

extern "C" int swiftk_get_tuning(int key) {
    switch (key) {
        case 0: return g_variant;
        case 1: return g_group_m;
        case 2: return g_persist_wgs;
        case 3: return g_dbg;
        case 4: return g_attn_dbg;
        case 5: return g_fwd_tiled;
        case 6: return g_modnorm_nt;
        case 7: return g_stagger_permille;
        case 8: return g_fwd_fused;
        case 9: return g_attn_bwd_pipe;
        case 11: return g_x3_exact;
        case 12: return g_fwd_pair;
        case 13: return g_f32_chunk_k;
        case 14: return g_fwd_splitk;
        case 15: return g_attn_bwd_fuse;
        case 16: return g_modnorm_bwd_fused;
        case 17: return g_modnorm_jvp_rows;
        case 18: return g_x3_ffsplit;
        case 19: return g_fwd_pepair;
        case 20: return g_pp;
        case 21: return g_attn_pp;
        case 22: return g_tn_pp;
        case 23: return g_fwd_rownorm;
        case 25: return g_zero_memset;
        case 26: return g_x3_normsplit;
        case 27: return g_x3_qkonly;
        case 28: return g_x3_attnpv;
        case 29: return g_fwd_tail;
    }
    return SWIFTK_EINVAL;
}

extern "C" int swiftk_profile_gemm(int epilogue, int64_t N) {
    g_prof.epilogue = epilogue;
    g_prof.N = (int)N;
    g_prof.used = 0;
    return 0;
}

extern "C" int swiftk_profile_collect(double* total_ms, int64_t* launches) {
    if (!total_ms || !launches) return SWIFTK_EINVAL;
    double tot = 0.0;
    for (int i = 0; i < g_prof.used; ++i) {
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return SWIFTK_EINVAL;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return SWIFTK_EINVAL;
        tot += ms;
    }
    *total_ms = tot;
    *launches = g_prof.used;
    g_prof.used = 0;
    return 0;
}

extern "C" int64_t swiftk_gemm_k_pad(int dtype, int64_t k) {
    const int64_t g = dtype == SWIFTK_BF16 ? 64 : 32;
    return (k + g - 1) / g * g;
}

static int gemm_impl(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                     int64_t K, int dtype, int out_dtype, int epilogue, const float* ep0, const float* ep1, int64_t pos_rows,
                     int ksplit, int64_t c_split, void* stream, const int* tiling = nullptr, float* kscr = nullptr,
                     int kchunk = 0, int64_t* tail_rows_from = nullptr) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || ksplit < 1) return SWIFTK_EINVAL;
    if (tail_rows_from && (ksplit != 1 || dtype != SWIFTK_BF16 || out_dtype != SWIFTK_BF16 || epilogue != SWIFTK_EPI_NONE || c_split < M * ldc))
        return SWIFTK_EINVAL;
    if (ksplit > 1 && ((out_dtype != SWIFTK_F32 && !(out_dtype == SWIFTK_BF16 && dtype == SWIFTK_BF16)) || epilogue != SWIFTK_EPI_NONE || c_split < M * ldc))
        return SWIFTK_EINVAL;
    if (epilogue == SWIFTK_EPI_ACCUM && out_dtype != SWIFTK_F32) return SWIFTK_EINVAL;
    if (epilogue == SWIFTK_EPI_SWIGLU_BWD && (out_dtype != SWIFTK_BF16 || !ep1 || ((uintptr_t)ep1 & 15) || pos_rows < 2 * N || pos_rows % 8 ||
                                              ldc < 2 * N || ldc % 8 || ((uintptr_t)C & 15)))
        return SWIFTK_EINVAL;
    if (epilogue == SWIFTK_EPI_SWIGLU_SPLIT3 && (dtype != SWIFTK_BF16 || out_dtype != SWIFTK_BF16 || pos_rows < N / 2 || ldc < 2 * pos_rows + N / 2))
        return SWIFTK_EINVAL;
    if (epilogue == SWIFTK_EPI_SWIGLU_BOTH && (out_dtype != SWIFTK_BF16 || !ep1 || ((uintptr_t)ep1 & 15) || pos_rows < N / 2 || pos_rows % 8 || N % 16))
        return SWIFTK_EINVAL;
    if (dtype != SWIFTK_F32 && dtype != SWIFTK_BF16) return SWIFTK_EINVAL;
    if (out_dtype != SWIFTK_F32 && out_dtype != dtype) return SWIFTK_EINVAL;
    const int es = dtype == SWIFTK_BF16 ? 2 : 4, os = out_dtype == SWIFTK_BF16 ? 2 : 4;
    // K must fill whole 128-B k-tiles, or end exactly half-way into the last one provided both operands' rows extend
    // (zero- / finitely-padded) to the end of that tile; the persistent kernel then skips the empty half.
    const int tile_k = ROWB / es;
    int khalf = 0;
    if (K % tile_k == tile_k / 2 && lda >= K + tile_k / 2 && ldw >= K + tile_k / 2) {
        khalf = 1;
        K += tile_k / 2;
    }
    if (K % tile_k != 0 || N % 4 != 0) return SWIFTK_ESHAPE;
    if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return SWIFTK_ESHAPE;
    if (lda < K || ldw < K) return SWIFTK_ESHAPE;
    const int ovec = (epilogue == SWIFTK_EPI_SWIGLU || epilogue == SWIFTK_EPI_SWIGLU_SPLIT3 ? 2 : 4) * os;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || (lda * es) % 16 || (ldw * es) % 16) return SWIFTK_EALIGN;
    if (((uintptr_t)C % ovec) || (ldc * os) % ovec) return SWIFTK_EALIGN;
    // QKNORM: whole heads of 3 x head_dim columns, a wave tile = two head vectors -> tile width 4 x head_dim; head_dim
    // travels in `pos_rows` (0 = 88); fp32 operands: whole tiles only for 80 / 96 (M, N multiples of 8)
    int ni = 11;
    int qk_only = 0;
    if (epilogue == SWIFTK_EPI_QKNORM && pos_rows < 0) {  // [q | k] pairs only (see the header): head_dim = -pos_rows
        qk_only = 1;
        pos_rows = -pos_rows;
        if (dtype != SWIFTK_F32 || out_dtype != SWIFTK_F32 || N % (4 * pos_rows) != 0) return SWIFTK_ESHAPE;
    }
    if (epilogue == SWIFTK_EPI_QKNORM) {
        const int64_t hd = pos_rows > 0 ? pos_rows : 88;
        // whole head pairs (a wave tile is two head vectors); fp32 operands also take whole single heads (N % 3 head_dim: the split
        // engine recomputes ONE hot head, whose [q | k | v] is one tile column with its fourth vector empty)
        const int64_t unit = qk_only ? 4 * hd : (dtype == SWIFTK_F32 && out_dtype == SWIFTK_F32 ? 3 * hd : 6 * hd);
        if (!ep0 || (hd != 80 && hd != 88 && hd != 96) || N % unit != 0) return SWIFTK_ESHAPE;
        if (hd != 88 && dtype != SWIFTK_BF16 && (g_variant == 0 || (M & 7) || (N & 7))) return SWIFTK_ESHAPE;
        ni = (int)(hd / 8);
    } else if (dtype == SWIFTK_BF16 && N % 352 != 0) {  // tile width that divides N, if one does (dim 1280 / 1536 families)
        if (N % 384 == 0) ni = 12;
        else if (N % 320 == 0) ni = 10;
    }
    if (epilogue == SWIFTK_EPI_BIAS_POS && (!ep0 || ((uintptr_t)ep0 & 15) || (ep1 && (((uintptr_t)ep1 & 15) || pos_rows <= 0))))
        return SWIFTK_EINVAL;
    GemmArgs g;
    g.A = static_cast<const char*>(A);
    g.W = static_cast<const char*>(W);
    g.C = static_cast<char*>(C);
    g.lda_b = lda * es;
    g.ldw_b = ldw * es;
    g.ldc = ldc;
    g.M = (int)M;
    g.N = (int)N;
    g.K = (int)K;
    g.ep0 = ep0;
    g.ep1 = ep1;
    g.pos_rows = (int)pos_rows;
    g.ni = ni;
    g.qk_only = qk_only;
    g.ntn = (int)((N + 32 * ni - 1) / (32 * ni));
    g.dbg = g_dbg;
    g.ksplit = ksplit;
    g.c_split = c_split;
    g.batch_a = g.batch_w = g.batch_c = 0;
    g.khalf = khalf;
    g.touch = 0;  // (unused: the look-ahead is a build-time switch, its requests clamp against the matrix edges)
    // estimated time of one output tile at ~1.35 PFLOP/s chip-wide (5.3 TFLOP/s per CU), in 10-ns ticks; the stagger only
    // makes sense when a workgroup walks several tiles
    g.stagger = 0;
    if (g_stagger_permille > 0 && dtype == SWIFTK_BF16 && ksplit == 1) {
        const double tile_s = 2.0 * 256.0 * (32.0 * ni) * (double)K / 5.3e12;
        const int64_t tiles = ((M + 255) / 256) * g.ntn;
        const int pm = g_stagger_permille % 10000;  // key 7 values >= 10000: per-XCD phases
        if (tiles >= 4 * 256) g.stagger = (int)(tile_s * 1e8 / 8.0 * pm / 1000.0) * (g_stagger_permille >= 10000 ? -1 : 1);
    }
    g.kscr = kscr;
    g.kchunk = 0;
    if (kscr && dtype == SWIFTK_F32 && ksplit == 1 && kchunk > 0) {
        // chains of equal length: K = 1056 (33 k-tiles) at a nominal 8 tiles -> 4 chains of 9, 9, 9, 6 (three parks per tile),
        // not 8, 8, 8, 8, 1
        const int nkt = (int)(K / tile_k), chains = nkt / kchunk > 1 ? nkt / kchunk : 1;
        g.kchunk = chains > 1 ? (nkt + chains - 1) / chains : 0;
    }
    g.t_gh = g.t_gw = g.t_sh = g.t_sw = g.t_heads = 0;
    if (tiling) {
        g.t_gh = tiling[0]; g.t_gw = tiling[1]; g.t_sh = tiling[2]; g.t_sw = tiling[3]; g.t_heads = tiling[4];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    g.tail_from = 0;
    if (tail_rows_from) {
        // The last round of the persistent walk as k-halves (see EPI_NONE_TAIL in the kernel): T tiles on G workgroups with a
        // remainder r = T % G of at most G / 2 -- the first T - r tiles whole into slab 0, the last r as two halves into slabs 0 / 1.
        // The caller's norm gets the walk's description (tail[3] = first row of the tile group the first split tile sits in, the first
        // split tile, the group height): it reads slab 1 exactly under the split tiles; slab 1 is not written anywhere else.
        const int G = g_persist_wgs, ntm = (int)((M + BM - 1) / BM);
        const int64_t T = (int64_t)ntm * g.ntn, r = T % G;
        const int nk_all = (int)(K / tile_k);
        if (ni != NI || (M & 7) || (N & 7) || ((uintptr_t)C & 15) || (ldc & 7) || (c_split & 7) || g_variant == 0 || !pp_ok(g) || nk_all < 6 ||
            T <= G || r == 0 || 2 * r > G)
            return SWIFTK_ESHAPE;
        g.tail_from = (int)(T - r);
        const int64_t per = (int64_t)g_group_m * g.ntn, g0 = g.tail_from / per;
        const int64_t row0 = g0 * g_group_m * BM;
        tail_rows_from[0] = row0 < M ? row0 : M;
        tail_rows_from[1] = g.tail_from;
        tail_rows_from[2] = g_group_m;
        const int items = (int)(2 * T - g.tail_from);
        hipLaunchKernelGGL((gemm_kernel_p<bf16_t, bf16_t, EPI_NONE_TAIL, NI, true>), dim3(items < G ? items : G), dim3(NT), 0, st, g, ntm, g_group_m);
        SWIFTK_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == SWIFTK_BF16) {
        if (out_dtype == SWIFTK_BF16) return dispatch_epi<bf16_t, bf16_t>(epilogue, g, st);
        return dispatch_epi<bf16_t, float>(epilogue, g, st);
    }
    return dispatch_epi<float, float>(epilogue, g, st);
}

extern "C" int swiftk_gemm_tail_split_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* slabs, int64_t ldc,
                                           int64_t slab_stride, int64_t M, int64_t N, int64_t K, int64_t* tail, void* stream) {
    if (!tail) return SWIFTK_EINVAL;
    return gemm_impl(A, lda, W, ldw, slabs, ldc, M, N, K, SWIFTK_BF16, SWIFTK_BF16, SWIFTK_EPI_NONE, nullptr, nullptr, 0, 1, slab_stride,
                     stream, nullptr, nullptr, 0, tail);
}

extern "C" int swiftk_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                           int64_t N, int64_t K, int dtype, int out_dtype, int epilogue, const float* ep0, const float* ep1,
                           int64_t pos_rows, void* stream) {
    return gemm_impl(A, lda, W, ldw, C, ldc, M, N, K, dtype, out_dtype, epilogue, ep0, ep1, pos_rows, 1, 0, stream);
}

extern "C" int swiftk_gemm_jvp(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t Mh, int64_t N,
                               int64_t K, int epilogue, const float* scale, float* rn, int head_dim, void* C2, int64_t ldc2,
                               void* stream) {
    if (!A || !W || Mh <= 0 || N <= 0 || K <= 0) return SWIFTK_EINVAL;
    if (epilogue != SWIFTK_EPI_QKNORM_JVP && epilogue != SWIFTK_EPI_SWIGLU_JVP) return SWIFTK_EINVAL;
    if (Mh % 128 || 2 * Mh > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return SWIFTK_ESHAPE;
    int khalf = 0;
    if (K % 64 == 32 && lda >= K + 32 && ldw >= K + 32) {
        khalf = 1;
        K += 32;
    }
    if (K % 64 || lda < K || ldw < K) return SWIFTK_ESHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || (lda * 2) % 16 || (ldw * 2) % 16) return SWIFTK_EALIGN;
    int ni = 11;
    if (epilogue == SWIFTK_EPI_QKNORM_JVP) {
        const int hd = head_dim > 0 ? head_dim : 88;
        if (!C || !scale || (hd != 80 && hd != 88 && hd != 96) || N % (6 * hd) || ldc < N) return SWIFTK_ESHAPE;
        if (((uintptr_t)C & 15) || ldc % 8) return SWIFTK_EALIGN;
        ni = hd / 8;
    } else {
        if (!C2 || N % 16 || ldc2 < N / 2 || (C && ldc < N)) return SWIFTK_ESHAPE;
        if (((uintptr_t)C2 & 15) || ldc2 % 8 || (C && (((uintptr_t)C & 15) || ldc % 8))) return SWIFTK_EALIGN;
        if (N % 384 == 0 && N % 352) ni = 12;
        else if (N % 320 == 0 && N % 352) ni = 10;
    }
    GemmArgs g;
    g.A = static_cast<const char*>(A);
    g.W = static_cast<const char*>(W);
    g.C = static_cast<char*>(C);
    g.lda_b = lda * 2;
    g.ldw_b = ldw * 2;
    g.ldc = ldc;
    g.M = (int)(2 * Mh);
    g.N = (int)N;
    g.K = (int)K;
    g.ep0 = scale;
    g.ep1 = epilogue == SWIFTK_EPI_QKNORM_JVP ? rn : static_cast<const float*>(C2);
    g.pos_rows = (int)ldc2;
    g.ni = ni;
    g.ntn = (int)((N + 32 * ni - 1) / (32 * ni));
    g.dbg = g_dbg;
    g.ksplit = 1;
    g.c_split = 0;
    g.batch_a = g.batch_w = g.batch_c = 0;
    g.khalf = khalf;
    g.touch = 0;
    g.stagger = 0;
    g.kscr = nullptr;
    g.kchunk = 0;
    g.t_gh = g.t_gw = g.t_sh = g.t_sw = g.t_heads = 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (epilogue == SWIFTK_EPI_QKNORM_JVP) return launch_paired<SWIFTK_EPI_QKNORM_JVP>(g, st);
    return launch_paired<SWIFTK_EPI_SWIGLU_JVP>(g, st);
}

extern "C" int swiftk_gemm_bias_pos_pair(const void* A, int64_t lda, const void* W, int64_t ldw, void* hi, int64_t ldh, void* lo,
                                         int64_t ldl, int64_t M, int64_t N, int64_t K, const float* bias, const float* pos,
                                         int64_t pos_rows, void* stream) {
    if (!A || !W || !hi || !lo || !bias || M <= 0 || N <= 0 || K <= 0 || (pos && pos_rows <= 0)) return SWIFTK_EINVAL;
    if (M % 8 || N % 16 || K % 64 || lda < K || ldw < K || ldh < N || ldl < N || M > (1 << 30) || N > (1 << 30)) return SWIFTK_ESHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)hi & 15) || ((uintptr_t)lo & 15) || ((uintptr_t)bias & 15) ||
        ((uintptr_t)pos & 15) || (lda * 2) % 16 || (ldw * 2) % 16 || ldh % 8 || ldl % 16)
        return SWIFTK_EALIGN;
    // 320-wide tiles where they divide N (dim 1280); dim 1536 takes the 352-wide ones with a partly filled last column (launch_paired)
    const int ni = N % 352 && N % 320 == 0 ? 10 : 11;
    GemmArgs g;
    g.A = static_cast<const char*>(A);
    g.W = static_cast<const char*>(W);
    g.C = static_cast<char*>(hi);
    g.lda_b = lda * 2;
    g.ldw_b = ldw * 2;
    g.ldc = ldh;
    g.M = (int)M;
    g.N = (int)N;
    g.K = (int)K;
    g.ep0 = bias;
    g.ep1 = pos;
    g.pos_rows = (int)pos_rows;
    g.ni = ni;
    g.ntn = (int)((N + 32 * ni - 1) / (32 * ni));
    g.dbg = g_dbg;
    g.ksplit = 1;
    g.c_split = ldl;  // (row stride of the low parts, bytes)
    g.batch_a = g.batch_w = g.batch_c = 0;
    g.khalf = 0;
    g.touch = 0;
    g.stagger = 0;
    g.kscr = static_cast<float*>(lo);
    g.kchunk = 0;
    g.t_gh = g.t_gw = g.t_sh = g.t_sw = g.t_heads = 0;
    return launch_paired<EPI_BIAS_POS_PAIR>(g, static_cast<hipStream_t>(stream));
}

extern "C" int64_t swiftk_gemm_chunk_scratch_bytes(void) {
    // persistent grid x 8 waves x (4 x 12) accumulator tiles x 64 lanes x 16 B (the widest tile geometry)
    return (int64_t)(g_persist_wgs > 256 ? g_persist_wgs : 256) * 8 * (MI * 12) * 64 * 16;
}

extern "C" int swiftk_gemm_chunked(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                   int64_t N, int64_t K, int dtype, int out_dtype, int epilogue, const float* ep0,
                                   const float* ep1, int64_t pos_rows, int chunk_k, void* scratch, int64_t scratch_bytes,
                                   void* stream) {
    if (dtype != SWIFTK_F32 || chunk_k <= 0 || chunk_k % 32) return SWIFTK_EINVAL;
    if (!scratch || ((uintptr_t)scratch & 15)) return SWIFTK_EALIGN;
    if (scratch_bytes < swiftk_gemm_chunk_scratch_bytes()) return SWIFTK_EWORKSPACE;
    return gemm_impl(A, lda, W, ldw, C, ldc, M, N, K, dtype, out_dtype, epilogue, ep0, ep1, pos_rows, 1, 0, stream, nullptr,
                     static_cast<float*>(scratch), chunk_k / 32);
}

// C_b = A_b W_b^T for b < batch, every matrix of a stack at a constant step from the previous one: one launch of the
// one-tile-per-workgroup kernel with blockIdx.y = b.  For many SMALL products (the Newton-Schulz iterations of Muon over the
// twelve same-shape weights of a layer class: 15 output tiles each) the batch fills the chip where a single product needs
// split-K slabs and a reduction pass to do so.
extern "C" int swiftk_gemm_batched(const void* A, int64_t lda, int64_t stride_a, const void* W, int64_t ldw, int64_t stride_w, void* C,
                                   int64_t ldc, int64_t stride_c, int batch, int64_t M, int64_t N, int64_t K, int dtype,
                                   int out_dtype, void* stream) {
    if (!A || !W || !C || batch <= 0 || batch > 65535 || M <= 0 || N <= 0 || K <= 0) return SWIFTK_EINVAL;
    if (dtype != SWIFTK_BF16 || (out_dtype != SWIFTK_BF16 && out_dtype != SWIFTK_F32)) return SWIFTK_EINVAL;
    const int os = out_dtype == SWIFTK_BF16 ? 2 : 4;
    if (K % 64 || N % 4 || lda < K || ldw < K || ldc < N) return SWIFTK_ESHAPE;
    if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return SWIFTK_ESHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || (lda * 2) % 16 || (ldw * 2) % 16 || (stride_a * 2) % 16 || (stride_w * 2) % 16)
        return SWIFTK_EALIGN;
    if (((uintptr_t)C % (4 * os)) || (ldc * os) % (4 * os) || (stride_c * os) % (4 * os)) return SWIFTK_EALIGN;
    GemmArgs g;
    g.A = static_cast<const char*>(A);
    g.W = static_cast<const char*>(W);
    g.C = static_cast<char*>(C);
    g.lda_b = lda * 2;
    g.ldw_b = ldw * 2;
    g.ldc = ldc;
    g.M = (int)M;
    g.N = (int)N;
    g.K = (int)K;
    g.ep0 = g.ep1 = nullptr;
    g.pos_rows = 0;
    g.ni = 11;
    g.ntn = (int)((N + BN - 1) / BN);
    g.dbg = 0;
    g.khalf = g.touch = g.stagger = 0;
    g.ksplit = 1;
    g.c_split = 0;
    g.kscr = nullptr;
    g.kchunk = 0;
    g.batch_a = stride_a * 2;
    g.batch_w = stride_w * 2;
    g.batch_c = stride_c * os;
    g.t_gh = g.t_gw = g.t_sh = g.t_sw = g.t_heads = 0;
    const dim3 grid((unsigned)(((M + BM - 1) / BM) * g.ntn), (unsigned)batch);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out_dtype == SWIFTK_BF16) hipLaunchKernelGGL((gemm_kernel<bf16_t, bf16_t, SWIFTK_EPI_NONE>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<bf16_t, float, SWIFTK_EPI_NONE>), grid, dim3(NT), 0, st, g);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_gemm_qkv_tiled(const void* A, int64_t lda, const void* W, int64_t ldw, void* qkv_tiled, int64_t K,
                                     const float* scale, int B, int gh, int gw, int heads, int head_dim, int shift_h,
                                     int shift_w, void* stream) {
    if (B <= 0 || heads <= 0 || gh <= 0 || gw <= 0 || gh % 16 || gw % 16) return SWIFTK_ESHAPE;
    if (head_dim != 80 && head_dim != 88 && head_dim != 96) return SWIFTK_ESHAPE;
    if (shift_h < 0 || shift_w < 0 || shift_h >= gh || shift_w >= gw) return SWIFTK_ESHAPE;
    if ((uintptr_t)qkv_tiled & 15) return SWIFTK_EALIGN;
    if (g_variant == 0) return SWIFTK_ESHAPE;  // the tiled store lives in the persistent kernel's epilogue
    const int tiling[5] = {gh, gw, shift_h, shift_w, heads};
    const int64_t M = (int64_t)B * gh * gw, N = 3 * (int64_t)heads * head_dim;
    return gemm_impl(A, lda, W, ldw, qkv_tiled, N, M, N, K, SWIFTK_BF16, SWIFTK_BF16, SWIFTK_EPI_QKNORM, scale, nullptr,
                     head_dim, 1, 0, stream, tiling);
}

// the same with bf16 slabs (each partial product rounded once, as a plain bf16 GEMM rounds its output): the one-unit-per-step form
// of wo / w2, whose consumer (swiftk_modnorm_residual_pair_slabs_bf16) then reads 4 instead of 8 bytes per element of y
extern "C" int swiftk_gemm_splitk_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* slabs, int64_t ldc,
                                       int64_t slab_stride, int64_t M, int64_t N, int64_t K, int ksplit, void* stream) {
    if ((M & 7) || (N & 7) || ((uintptr_t)slabs & 15) || (ldc & 7) || (slab_stride & 7)) return SWIFTK_ESHAPE;  // persistent kernel only
    return gemm_impl(A, lda, W, ldw, slabs, ldc, M, N, K, SWIFTK_BF16, SWIFTK_BF16, SWIFTK_EPI_NONE, nullptr, nullptr, 0, ksplit,
                     slab_stride, stream);
}

extern "C" int swiftk_gemm_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, float* slabs, int64_t ldc,
                                  int64_t slab_stride, int64_t M, int64_t N, int64_t K, int dtype, int ksplit, void* stream) {
    return gemm_impl(A, lda, W, ldw, slabs, ldc, M, N, K, dtype, SWIFTK_F32, SWIFTK_EPI_NONE, nullptr, nullptr, 0, ksplit,
                     slab_stride, stream);
}
