// to_qkv + cosine-norm + shifted-window attention in ONE kernel for gfx950 (bf16, head_dim 80 / 88 / 96, 16x16 windows).
//
// Replaces reference src/swift/models/swinv2.py:119-136 + 185-208 for one layer: F.linear(to_qkv), the per-head
// [q|k|v] split, q/k L2-normalisation and logit scale, roll + window_partition, softmax(q k^T) v, window_reverse + roll.
// The separate kernels (swiftk_gemm_qkv_tiled + swiftk_window_attention) move q, k, v through HBM once each way:
// 8192 x 3168 bf16 written and read back per sample and layer (104 MB, 10 GB per layer at 96 units).  Here a work item is
// one (sample, window, head): its 256 x 264 slab of q|k|v is produced by an MFMA GEMM over the window's 256 token rows
// (gathered in window order by the LDS-DMA source addresses, roll included) and the head's 264 weight rows, normalised on
// the fp32 accumulators, parked as bf16 in the LDS the k-loop has just released, and consumed by the attention core on
// the spot.  Only the 256 x 88 output tile leaves the CU.
//
// Geometry: 8 waves as 4 (M) x 2 (N); a wave owns 64 token rows x 128 or 144 columns = 4 x 8 / 4 x 9 MFMA 16x16 tiles (128 /
// 144 accumulator VGPRs), v_mfma_f32_16x16x32_bf16 with the roles of gemm.hip (A := weight rows, B := token rows -> a lane
// holds 4 consecutive columns of one row).  The slab's columns are ORDERED for that split by the weight rows' DMA source
// addresses: column half 0 = [q (88) | v 0..39], half 1 = [k (88) | v 40..87 | 8 pad] -- so a wave holds complete q rows or
// complete k rows of its 64 tokens and the cosine norms need no cross-wave exchange (v has no norm); the 8 pad columns are
// fed by a clamped weight row and never stored.  Waves wv and wv + 4 (partners on a SIMD) take the two halves of the same
// rows, so every SIMD carries 8 + 9 column blocks; the k-loop and the norm / store epilogue are instantiated once per
// block count and selected per wave OUTSIDE the loops (a per-wave trip count inside the MFMA stream is a branch there, and
// hipcc answers branches with split accumulator tuples).  13 LDS fragment reads per 36 MFMAs against 19 per 34 for a wave
// that spans all columns.  k-loop: 64-deep k-tiles, two LDS
// stages filled by global_load_lds_dwordx4 (1 KiB pieces of 8 rows x 128 B, source-side XOR swizzle), one barrier per
// k-tile -- the structure of gemm_kernel_p.  Attention core: the S^T = K Q^T / accumulator-as-operand / V^T-by-
// ds_read_tr16 scheme of attention_pipe.hip (v_mfma_f32_32x32x16_bf16, a wave owns 32 queries, softmax streamed over four
// 64-key chunks, no row maximum where exp(min(scale, ln 100)) <= 48, row sum on the spare V^T rows).
//
// LDS (161,792 B):  [0, 45056) K tile | [45056, 94208) V tile (192-B rows) | [94208, 139264) Q tile | tail to 161,792
//   k-loop stage 1 (odd k-tiles)  = bytes [0, 67584)          (over the K / V tiles, dead during the k-loop)
//   k-loop stage 0 (even k-tiles) = bytes [94208, 161792)     (over the Q tile and the tail)
// K = 1056 is 16.5 k-tiles: the last one (index 16, stage 0) carries data in its first half only.  After it: barrier, the
// normalised q / k / v slabs are written over both stages, barrier, every wave pulls its Q fragments into registers,
// barrier, and the NEXT item's first k-tile is requested into stage 0 (the Q tile is dead by then) so that it lands
// under the attention core; the output tile leaves through wave-private slabs in the K tile once all waves are done with
// K and V.
#include <type_traits>

#include "common.h"

namespace {

constexpr int NT = 512;
constexpr int ROWB = 128;                   // bytes of a k-tile row (64 bf16)
constexpr int BM = 256;                     // tokens of a window
constexpr int MI = 4;
constexpr int A_BYTES = BM * ROWB;          // 32 KiB
constexpr int VROW = 192;                   // V rows are padded to 192 B: the transposed reads (ds_read_b64_tr_b16: 4 rows x 64 B
                                            // per 32-lane group) are 2-way bank conflicts on 176-B rows and conflict-free on 192
constexpr int VTILE = 256 * VROW;           // 49152
constexpr int CH = 64, NST = 4, DB = 3;
constexpr int OROWS = 16, ORND = 2;         // output staging: 16 rows per round, two rounds per item
constexpr int LDS_MAX = 163840;             // the CU's 160 KiB
constexpr float LOG2E = 1.4426950408889634f;

// Geometry per head_dim (80 / 88 / 96: Swift's 468 M variant, Swift-B, the 664 M variant -- era5-swinv2-1.4-scm.yaml:29-36).
// The slab's 3 HD columns are split into two column halves of whole 16-column MFMA blocks so that a wave holds complete q rows
// (half 0) or complete k rows (half 1) of its 64 tokens:  half 0 = [q (HD) | v 0 .. NV0-1],  half 1 = [k (HD) | v NV0 .. HD-1 | pad]
//   HD 80:  128 = 80 + 48   |  112 = 80 + 32        (NI 8 | 7, no pad)
//   HD 88:  128 = 88 + 40   |  144 = 88 + 48 + 8    (NI 8 | 9)
//   HD 96:  144 = 96 + 48   |  144 = 96 + 48        (NI 9 | 9, no pad)
template <int HD_>
struct Geo {
    static constexpr int HD = HD_;
    static constexpr int NI0 = HD == 96 ? 9 : 8;               // MFMA 16-column blocks per wave of column half 0
    static constexpr int WT0 = 16 * NI0;                       // columns of half 0
    static constexpr int NV0 = WT0 - HD, NV1 = HD - NV0;       // v columns of half 0 / half 1
    static constexpr int NI1 = (HD + NV1 + 15) / 16;           // blocks of half 1
    static constexpr int BNP = WT0 + 16 * NI1;                 // LDS image rows of the weight operand: 240 / 272 / 288
    static constexpr int W_BYTES = BNP * ROWB;
    static constexpr int STAGE = A_BYTES + W_BYTES;            // 63488 / 67584 / 69632
    static constexpr int ROW = HD * 2;                         // a q / k row (and a v row's data)
    static constexpr int TILE = 256 * ROW;                     // 40960 / 45056 / 49152
    static constexpr int OFF_K = 0, OFF_V = TILE, OFF_Q = TILE + VTILE;
    static constexpr int OFF_S1 = 0;
    // stage 0 lies over the Q tile and the tail.  head_dim 96: K + V + stage = 164 KiB -- stage 0 starts 4 KiB early, over the end
    // of the V tile; harmless inside the k-loop (V is dead there), and the four 1-KiB pieces that land there when the next item's
    // first k-tile is requested under the attention core (wave 0's token pieces) are requested behind the core instead (DEFER)
    static constexpr int OFF_S0 = OFF_Q + STAGE <= LDS_MAX ? OFF_Q : LDS_MAX - STAGE;
    static constexpr bool DEFER = OFF_S0 < OFF_Q;
    static constexpr int LDS_TOTAL = OFF_S0 + STAGE;
    static constexpr int CHB = CH * ROW;
    static constexpr int KS = (HD + 15) / 16;                  // 16-deep steps of S^T = K Q^T
    static constexpr bool KS_HALF = HD % 16 != 0;              // the last step has one real 8-element chunk (head_dim 88)
    static constexpr bool ONES = HD < 32 * DB;                 // spare V^T rows carry ones: the row sum rides on the matrix pipe
    static constexpr int CPR = HD / 8;                         // 16-B chunks of an output row
    static constexpr int OSLAB = OROWS * ROW;
    static constexpr int NOST = ORND * ((OROWS * CPR + 63) / 64);  // output store instructions per wave and item
    static constexpr int WP = W_BYTES / 1024;                  // weight pieces per stage: 30 / 34 / 36
    static_assert(OFF_S1 + STAGE <= OFF_Q, "stage 1 must fit over the K and V tiles");
    static_assert(OFF_S0 + STAGE <= LDS_MAX && OFF_Q + TILE <= LDS_MAX, "LDS plan");
    static_assert(!DEFER || OFF_Q - OFF_S0 <= 4096, "only wave 0's four token pieces may land in the overlap");
};

struct FusedArgs {
    const char* x;      // [B * gh * gw, ldx] bf16 token-major operand copy of the residual stream
    const char* w;      // [3 * heads * 88, ldw] bf16, per-head [q|k|v] row interleave (to_qkv.weight as stored)
    bf16_t* out;        // [B * gh * gw, ldo] bf16, head h in columns [88 h, 88 h + 88)
    const float* scale;  // [heads] logit scale parameter
    int64_t ldx_b, ldw_b, ldo;
    int B, gh, gw, heads, sh, sw;
    int nk, khalf;      // k-tiles, and whether the last one is half full
    int dbg;            // timing experiments: 1 = skip the attention core, 2 = every item reads head 0's weights, 4 = no parking writes, 8 = no norm + no parking writes
};

__device__ __forceinline__ int win_token(int wy, int wx, int j, int gh, int gw, int sh, int sw) {
    int gy = wy * 16 + (j >> 4) + sh;
    int gx = wx * 16 + (j & 15) + sw;
    gy = gy >= gh ? gy - gh : gy;
    gx = gx >= gw ? gx - gw : gx;
    return gy * gw + gx;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int HD, bool PP>
__global__ __launch_bounds__(NT) void qkv_attn_kernel(FusedArgs a, int nitems) {
    using G = Geo<HD>;
    constexpr int NI0 = G::NI0, NI1 = G::NI1, WT0 = G::WT0, NV0 = G::NV0, NV1 = G::NV1, BNP = G::BNP, STAGE = G::STAGE, ROW = G::ROW;
    constexpr int OFF_K = G::OFF_K, OFF_V = G::OFF_V, OFF_Q = G::OFF_Q, OFF_S0 = G::OFF_S0, OFF_S1 = G::OFF_S1, LDS_TOTAL = G::LDS_TOTAL;
    constexpr int CHB = G::CHB, KS = G::KS, CPR = G::CPR, OSLAB = G::OSLAB, NOST = G::NOST, WP = G::WP;
    __shared__ __attribute__((aligned(16))) char smem[LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwx = a.gw / 16, nw = (a.gh / 16) * nwx;
    const int64_t ntok = (int64_t)a.gh * a.gw;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

    // the workgroups of one XCD take consecutive items at the same time, each XCD walking its own contiguous eighth of the
    // item list in rounds (item numbering: `decode` below)
    int first, last, istep;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, nx = gridDim.x >> 3;
        first = (int)((int64_t)xcd * nitems / 8) + (blockIdx.x >> 3);
        last = (int)((int64_t)(xcd + 1) * nitems / 8);
        istep = nx;
    } else {
        first = (int)((int64_t)blockIdx.x * nitems / gridDim.x);
        last = (int)((int64_t)(blockIdx.x + 1) * nitems / gridDim.x);
        istep = 1;
    }
    if (first >= last) return;
    for (int o = tid * 16; o < LDS_TOTAL; o += NT * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // Items are numbered (sample, window, head), heads fastest: the 32 workgroups of an XCD work on the twelve heads of two
    // to three windows at a time, so a window's token rows are fetched into that XCD's L2 once and the twelve 176-B head
    // slices of an output row are completed in L2 by neighbouring CUs.  The twelve weight slabs (6.9 MB) do not fit the
    // 4 MB L2 beside them and stream from the Infinity Cache instead (PMC: 20 GB of fabric reads per launch at 96 units).
    // The opposite order -- groups of four heads outermost, so that four slabs stay L2-resident and the token rows are
    // fetched three times -- was measured 2 % slower (profiles/r03e_qkv_attn_fused_ab8.txt).
    auto decode = [&](int item, int& b, int& w, int& h) {
        h = item % a.heads;
        const int r = item / a.heads;
        w = r % nw;
        b = r / nw;
    };
    auto pin = [&](const char* base) {  // a provably wave-uniform pointer (SGPR pair) for the asm operand
        const uint64_t u = (uint64_t)base;
        return (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(u >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)u));
    };

    // ---- LDS-DMA sources.  A piece is 8 rows x 128 B, lane l -> row l >> 3, physical 16-B chunk l & 7, which holds the
    // row's logical chunk (l & 7) ^ ((row >> 1) & 7) (the fragment reads apply the same XOR: conflict-free ds_read_b128).
    // Token operand: this wave's pieces 4 wv .. 4 wv + 3 = window rows 32 wv .. 32 wv + 31 (its own query rows, as it
    // happens); the per-lane offset carries the gather (window partition of the grid rolled by (-sh, -sw)).
    // Weight operand: pieces wv + 8 i of the 272-row LDS image, i = 0..4 (the fifth exists for waves 0 and 1); image row r
    // takes the head's weight row  r (q) | 176 + r - 88 (v 0..39) | 88 + r - 128 (k) | r (v 40..87 = rows 216..263) | pad.
    const int prow = lane >> 3, pchunk = lane & 7;
    uint32_t va[4], vb[5];
    int cur_w = -1;
    auto set_window = [&](int w) {
        const int wy = w / nwx, wx = w - wy * nwx;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int j = (wv * 4 + p) * 8 + prow;
            va[p] = (uint32_t)(win_token(wy, wx, j, a.gh, a.gw, a.sh, a.sw) * (int)a.ldx_b) +
                    16u * (pchunk ^ ((4 * (p & 1) + (prow >> 1)) & 7));
        }
        cur_w = w;
    };
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int r = (wv + 8 * i) * 8 + prow;
        // image row r -> the head's weight row: q r | v (r - HD) | k (r - WT0) | v NV0 + (r - WT0 - HD) = row r again | pad (clamped)
        const int row = r < HD ? r : (r < WT0 ? 2 * HD + (r - HD) : (r < WT0 + HD ? HD + (r - WT0) : min(r, 3 * HD - 1)));
        vb[i] = (uint32_t)(row * (int)a.ldw_b) + 16u * (pchunk ^ ((4 * (wv & 1) + (prow >> 1)) & 7));
    }
    // Ping-pong k-loop (PP; the schedule of gemm.hip's persistent kernel): the weight image's rows form two regions -- LO = the
    // first four column blocks of both column halves (rows [0, 64) and [WT0, WT0 + 64): read in a k-half's first phase), HI = the
    // rest (second phase) -- and the weight pieces are dealt to the waves per region so that a wave can issue its HI pieces last
    // and wait for them one phase later than for everything else.  LO piece n = wv + 8 i < 16, HI piece n = wv + 8 i < NHI; the
    // piece's first image row is lo_row(i) / hi_row(i) (all region sizes are even piece counts: the swizzle parity stays wv & 1).
    constexpr int JA = 4, NH0 = (WT0 - 64) / 8, NHI = (BNP - 128) / 8;
    auto lo_row = [&](int i) { const int n = wv + 8 * i, hh_ = n >= 8; return hh_ * WT0 + (n - 8 * hh_) * 8; };
    auto hi_row = [&](int i) { const int n = wv + 8 * i, hh_ = n >= NH0; return hh_ ? WT0 + 64 + (n - NH0) * 8 : 64 + n * 8; };
    auto wrow = [&](int r) { return r < HD ? r : (r < WT0 ? 2 * HD + (r - HD) : (r < WT0 + HD ? HD + (r - WT0) : min(r, 3 * HD - 1))); };
    uint32_t vlo[2], vhi[3];
    if constexpr (PP) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            vlo[i] = (uint32_t)(wrow(lo_row(i) + prow) * (int)a.ldw_b) + 16u * (pchunk ^ ((4 * (wv & 1) + (prow >> 1)) & 7));
#pragma unroll
        for (int i = 0; i < 3; ++i)
            vhi[i] = (uint32_t)(wrow(min(hi_row(i), BNP - 8) + prow) * (int)a.ldw_b) + 16u * (pchunk ^ ((4 * (wv & 1) + (prow >> 1)) & 7));
    }
    const int c_hi = (wv < NHI - 16) ? 3 : ((wv + 8 < NHI) ? 2 : 1);  // HI pieces of this wave
    const char* xbase = nullptr;
    const char* wbase = nullptr;
    auto set_item = [&](int b, int w, int h) {
        if (w != cur_w) set_window(w);
        xbase = pin(a.x + (int64_t)b * ntok * a.ldx_b);
        // (dbg bit 2, timing experiment with WRONG results: every item reads head 0's weight slab -- 557 KB that stay in every
        // XCD's L2 -- to price the twelve slabs' streaming from the Infinity Cache: profiles/r05i_qkv_attn_w0.txt)
        wbase = pin(a.w + (int64_t)((a.dbg & 2) ? 0 : h) * (3 * HD) * a.ldw_b);
    };
    // piece p of a stage (0-3 token rows, 4-8 weight rows) for k-tile byte offset koff
    auto issue_piece = [&](uint32_t stage, uint32_t koff, int p) {
        if (p < 4) {
            dma_piece_fast(stage + (wv * 4 + p) * 1024, xbase, va[p] + koff);
        } else {
            const int i = p - 4;
            if (wv + 8 * i < WP) dma_piece_fast(stage + A_BYTES + (wv + 8 * i) * 1024, wbase, vb[i] + koff);
        }
    };

    auto pp_a = [&](uint32_t stage, uint32_t koff, int p_) { dma_piece_fast(stage + (wv * 4 + p_) * 1024, xbase, va[p_] + koff); };
    auto pp_lo = [&](uint32_t stage, uint32_t koff, int i) { dma_piece_fast(stage + A_BYTES + lo_row(i) * ROWB, wbase, vlo[i] + koff); };
    auto pp_hi = [&](uint32_t stage, uint32_t koff, int i) {
        if (wv + 8 * i < NHI) dma_piece_fast(stage + A_BYTES + hi_row(i) * ROWB, wbase, vhi[i] + koff);
    };
    // every piece of one k-tile (an item's first k-tile: requested ahead of the item, waited for in front of its k-loop);
    // `skip_a0`: head_dim 96 defers wave 0's token pieces behind the attention core (Geo::DEFER)
    auto issue_all = [&](uint32_t stage, uint32_t koff, bool skip_a) {
        if constexpr (PP) {
#pragma unroll
            for (int p_ = 0; p_ < 4; ++p_)
                if (!skip_a) pp_a(stage, koff, p_);
            pp_lo(stage, koff, 0);
            pp_lo(stage, koff, 1);
#pragma unroll
            for (int i = 0; i < 3; ++i) pp_hi(stage, koff, i);
        } else {
#pragma unroll
            for (int p_ = 0; p_ < 9; ++p_)
                if (!(skip_a && p_ < 4)) issue_piece(stage, koff, p_);
        }
    };

    // ---- fragment read offsets of the k-loop
    const int wm = wv & 3, wn = wv >> 2;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int xoff = (wm * 64 + r16) * ROWB;
    const int woff = A_BYTES + (wn * WT0 + r16) * ROWB;
    const int ch0 = ((g4 + 0) ^ (r16 >> 1)) * 16;
    const int ch1 = ((g4 + 4) ^ (r16 >> 1)) * 16;

    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);

    int b, w, h;
    decode(first, b, w, h);
    set_item(b, w, h);
    issue_all(lds0 + OFF_S0, 0u, false);
    bool have_prev = false;

    for (int item = first; item < last; item += istep) {
        int nb = b, nwn = w, nh = h;
        const bool has_next = item + istep < last;
        if (has_next) decode(item + istep, nb, nwn, nh);

        // =========================================================== k-loop: acc[i][j] = X_window W_head^T (fp32)
        auto gemm_part = [&](auto ni_tag) {
        constexpr int NI = decltype(ni_tag)::value;
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the first k-tile was requested before the previous item's output stores (VMEM retires in issue order)
        if (have_prev) wait_vm<NOST>(); else wait_vm<0>();
        if constexpr (PP) {
            // Ping-pong form (see gemm.hip, SWIFTK_X_PP): phases (k-half, column part) = (0, lo) (0, hi) (1, lo) (1, hi); in each a
            // wave first requests the phase's fragments and issues its share of the next stage's pieces (MEM), then -- behind a
            // barrier -- runs the phase's MFMAs back to back (COMPUTE), then a second barrier.  Waves 4-7 (column half 1), the SIMD
            // partners of waves 0-3, run one barrier behind.  Issue order per k-tile: A0 A1 A2 [wait HI of this k-tile] | A3 LO LO |
            // HI HI | HI [wait A, LO of the next k-tile, the HI pieces stay in flight].
            constexpr int JB = NI - JA;
            const int grp = wv >> 2;
            auto bar = [] {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            uint4 xf[MI], wf[JA > JB ? JA : JB];
            __builtin_amdgcn_s_barrier();  // the first stage landed for every wave
            if (grp) __builtin_amdgcn_s_barrier();
            for (int kt = 0; kt < a.nk; ++kt) {
                const char* s = smem + ((kt & 1) ? OFF_S1 : OFF_S0);
                const uint32_t fill = lds0 + ((kt & 1) ? OFF_S0 : OFF_S1);
                const bool more = kt + 1 < a.nk;
                const uint32_t koff = (uint32_t)(kt + 1) * ROWB;
                const bool half = a.khalf && !more;
                auto rdx = [&](const int ch) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const uint4*>(s + xoff + i * 16 * ROWB + ch);
                };
                auto rdw = [&](const int ch, const int j0, const int nj) {
#pragma unroll
                    for (int jj = 0; jj < (JA > JB ? JA : JB); ++jj)
                        if (jj < nj) wf[jj] = *reinterpret_cast<const uint4*>(s + woff + (j0 + jj) * 16 * ROWB + ch);
                };
                auto comp = [&](const int j0, const int nj) {
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int jj = 0; jj < (JA > JB ? JA : JB); ++jj) {
                        if (jj < nj) {
#pragma unroll
                            for (int i = 0; i < MI; ++i)
                                acc[i][j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[jj]),
                                                                                         __builtin_bit_cast(bf16x8, xf[i]), acc[i][j0 + jj], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                };
                // ---- (0, lo)
                rdx(ch0);
                rdw(ch0, 0, JA);
                if (more) {
                    pp_a(fill, koff, 0);
                    pp_a(fill, koff, 1);
                    pp_a(fill, koff, 2);
                    if (kt > 0) wait_vm<3>();   // this k-tile's HI pieces (the previous k-tile's last requests)
                } else if (kt > 0) {
                    wait_vm<0>();
                }
                bar();
                comp(0, JA);
                bar();
                // ---- (0, hi)
                rdw(ch0, JA, JB);
                if (more) {
                    pp_a(fill, koff, 3);
                    pp_lo(fill, koff, 0);
                    pp_lo(fill, koff, 1);
                }
                if (half) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage's last fragment reads
                bar();
                comp(JA, JB);
                bar();
                if (!half) {
                    // ---- (1, lo)
                    rdx(ch1);
                    rdw(ch1, 0, JA);
                    if (more) {
                        pp_hi(fill, koff, 0);
                        pp_hi(fill, koff, 1);
                    }
                    bar();
                    comp(0, JA);
                    bar();
                    // ---- (1, hi)
                    rdw(ch1, JA, JB);
                    if (more) {
                        pp_hi(fill, koff, 2);
                        if (c_hi == 3) wait_vm<3>(); else if (c_hi == 2) wait_vm<2>(); else wait_vm<1>();
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    bar();
                    comp(JA, JB);
                    bar();
                }
            }
            if (!grp) __builtin_amdgcn_s_barrier();
        } else
        for (int kt = 0; kt < a.nk; ++kt) {
            __builtin_amdgcn_s_barrier();  // stage of kt landed for every wave; every wave is done with the other stage
            const int so = (kt & 1) ? OFF_S1 : OFF_S0;
            const char* s = smem + so;
            const uint32_t fill = lds0 + ((kt & 1) ? OFF_S0 : OFF_S1);
            const bool more = kt + 1 < a.nk;
            // (one straight-line copy of the first k-half serves every k-tile -- a second, DMA-free copy for the last k-tile made
            // hipcc split and spill the accumulator tuples; the last k-tile skips its nine requests behind wave-uniform
            // branches instead, which cost no registers)
            const uint32_t koff = (uint32_t)(kt + 1) * ROWB;
            const bool half = a.khalf && !more;
            uint4 xf[MI];
            auto k_half = [&](const int ch, const bool with_dma) {
#pragma unroll
                for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const uint4*>(s + xoff + i * 16 * ROWB + ch);
                uint4 wf = *reinterpret_cast<const uint4*>(s + woff + ch);
                uint4 wf1 = *reinterpret_cast<const uint4*>(s + woff + 16 * ROWB + ch);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const uint4 wn_ = wf1;
                    if (j + 2 < NI) wf1 = *reinterpret_cast<const uint4*>(s + woff + (j + 2) * 16 * ROWB + ch);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf),
                                                                            __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
                    if (with_dma && more) issue_piece(fill, koff, j);
                    if (with_dma && more && j == NI - 1) {  // (nine pieces, seven to nine column blocks)
#pragma unroll
                        for (int p = NI; p < 9; ++p) issue_piece(fill, koff, p);
                    }
                    wf = wn_;
                }
            };
            k_half(ch0, true);
            if (!half) k_half(ch1, false);
            wait_vm<0>();
        }

        // =========================================================== epilogue: cosine norm on the accumulators -> LDS
        __builtin_amdgcn_s_barrier();  // every wave is done reading the last stage: both stages may be overwritten
        // Everything lane-derived below is rebuilt from an opaque copy of the lane id: left visible, hipcc hoists the
        // epilogue's, the attention core's and the output stage's address registers above the k-loop and spills inside it.
        int el = lane;
        asm volatile("" : "+v"(el));
        const int r16 = el & 15, g4 = el >> 4;
        {
            // wave (wm, wn): rows 64 wm .. + 63; local columns 0..87 = q (wn 0) or k (wn 1), 88.. = v 0..39 / v 40..87, then pad
            // (folding log2(e) into tau would save the core 128 v_mul per lane and item, but q-hat would then round differently
            // from the two-kernel path's: the two agree to 1e-4 as it is, 7e-3 with the fold -- kept as a regression check)
            const float tau = wn ? 1.0f : __expf(fminf(a.scale[h], 4.605170185988092f));
            const int nv = wn ? NV1 : NV0, v0 = wn ? NV0 : 0;
            char* qk_tile = smem + (wn ? OFF_K : OFF_Q);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (a.dbg & 8) continue;  // (timing probe, WRONG results: no norm arithmetic and no parking writes at all)
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < (HD + 15) / 16; ++j) {  // the head vector's blocks; head_dim 88 ends inside block 5 (lanes g4 < 2)
                    const f32x4 v = acc[i][j];
                    const float t = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    ss += (16 * j + 16 <= HD || 4 * g4 < HD - 16 * j) ? t : 0.f;
                }
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                const float f = tau / fmaxf(sqrtf(ss), 1e-12f);
                const int row = wm * 64 + i * 16 + r16;
                if (a.dbg & 4) continue;  // (timing probe, WRONG results: the k-loop -> core hand-off without its 135 KB of LDS writes)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int lc = 16 * j + 4 * g4;  // this lane's 4 columns of the wave's 144 (never straddle: 88 % 4 == 0)
                    const f32x4 v = acc[i][j];
                    if (lc < HD) {
                        *reinterpret_cast<uint2*>(qk_tile + row * ROW + lc * 2) =
                            make_uint2(pack_bf16(v[0] * f, v[1] * f), pack_bf16(v[2] * f, v[3] * f));
                    } else if (lc < HD + nv) {
                        // (V rows are 192 B = 48 banks apart, so the 16 rows of one ds_write_b64 fall on four bank groups.  Round 6
                        // XOR-ed the 8-B chunk index with (row >> 2) & 3 on both sides -- conflict-free writes, same read footprint,
                        // parity-green -- and measured +-0 (profiles/r06c_qkv_handoff_probe.txt): the hand-off is bound by the VALU
                        // issue of the two waves of a SIMD, not by LDS; the plain layout stays)
                        *reinterpret_cast<uint2*>(smem + OFF_V + row * VROW + (lc - HD + v0) * 2) =
                            make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                    }
                }
            }
        }
        };
        if (wn == 0) gemm_part(std::integral_constant<int, NI0>{}); else gemm_part(std::integral_constant<int, NI1>{});
        int el = lane;
        asm volatile("" : "+v"(el));
        const int c32 = el & 31, hh = el >> 5;
        const int vbase = (4 * hh + ((el & 15) >> 2)) * VROW + (16 * ((el >> 4) & 1) + 4 * (el & 3)) * 2;
        char* oslab = smem + OFF_K + wv * OSLAB;
        __builtin_amdgcn_s_barrier();  // Q, K and V tiles complete

        // =========================================================== attention core
        const float bound = __expf(fminf(a.scale[h], 4.605170185988092f));
        const bool online = !(bound <= 48.f);
        uint4 qf[KS];
        {
            const char* qrow = smem + OFF_Q + (wv * 32 + c32) * ROW;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (G::KS_HALF && ks == KS - 1) {  // head_dim 88 = 5.5 k-steps of 16: the last one has one real chunk
                    const uint4 t = *reinterpret_cast<const uint4*>(qrow + (2 * ks) * 16);
                    qf[ks] = hh ? make_uint4(0, 0, 0, 0) : t;
                } else {
                    qf[ks] = *reinterpret_cast<const uint4*>(qrow + (2 * ks + hh) * 16);
                }
            }
            // the compiler's own lgkmcnt wait precedes the first use; all waves must hold their fragments before the Q
            // tile is overwritten by the next item's first k-tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (has_next) {
            set_item(nb, nwn, nh);
            issue_all(lds0 + OFF_S0, 0u, G::DEFER && wv == 0);  // (head_dim 96: see Geo::OFF_S0)
        }
        f32x16 o[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
        // The four 64-key chunks one after the other (S^T, softmax, PV).  A software-pipelined form -- S^T of chunk t + 1 and PV
        // of chunk t - 1 issued beside the softmax of chunk t, with and without sched_group_barrier hints -- was measured
        // 2-3 % slower for the whole kernel (256 VGPRs and spills against 198); what overlaps the softmax's VALU work with
        // MFMAs is the SIMD's other wave, once the two are out of step (the stagger below).
        auto st_chunk = [&](int c, f32x16 (&sc)[2]) {
            const char* sK = smem + OFF_K + c * CHB;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[k2][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const uint4 kf = *reinterpret_cast<const uint4*>(sK + (k2 * 32 + c32) * ROW + ks * 32 + hh * 16);
                    sc[k2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                                     __builtin_bit_cast(bf16x8, qf[ks]), sc[k2], 0, 0, 0);
                }
            }
        };
        // (requesting a chunk's 24 transposed V^T reads in front of its softmax, so that the PV MFMAs issue back to back instead
        // of each behind an `lgkmcnt(0)` on reads issued just before it, was measured 2 % SLOWER for the kernel: 231 VGPRs)
        auto pv_chunk = [&](int c, const uint4 (&pf)[4]) {
            const char* sV = smem + OFF_V + c * (CH * VROW);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const char* vrow = sV + (k2 * 32 + s2 * 16) * VROW + vbase;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(vrow + db * 64));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(vrow + db * 64 + 8 * VROW));
                        uint4 vf = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                        if (G::ONES && db == DB - 1 && c32 >= HD - 32 * (DB - 1)) vf = ones;  // rows HD..95 of V^T: the row sum
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                                        __builtin_bit_cast(bf16x8, pf[2 * k2 + s2]), o[db], 0, 0, 0);
                    }
                }
        };
        float m_run = -INFINITY;
        float alpha_next = 1.f;  // online form: factor the softmax of step t found for O before PV of chunk t
        float l_valu = 0.f;      // head_dim 96 only: no spare V^T rows, the row sum is added up on the VALU
        auto soft_chunk = [&](const f32x16 (&sc)[2], uint4 (&pf)[4], auto online_tag) {
            float mb = 0.f;
            if constexpr (decltype(online_tag)::value) {
                float mx = m_run;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[k2][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                alpha_next = __builtin_amdgcn_exp2f((m_run - mx) * LOG2E);
                if constexpr (!G::ONES) l_valu *= alpha_next;
                m_run = mx;
                mb = mx * LOG2E;
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                float e[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(sc[k2][r] * LOG2E - mb);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    pf[2 * k2 + s2].x = pack_bf16(e[8 * s2 + 0], e[8 * s2 + 1]);
                    pf[2 * k2 + s2].y = pack_bf16(e[8 * s2 + 2], e[8 * s2 + 3]);
                    pf[2 * k2 + s2].z = pack_bf16(e[8 * s2 + 4], e[8 * s2 + 5]);
                    pf[2 * k2 + s2].w = pack_bf16(e[8 * s2 + 6], e[8 * s2 + 7]);
                    if constexpr (!G::ONES) {  // the sum of exactly the bf16-rounded probabilities the numerator uses
                        const uint32_t pw[4] = {pf[2 * k2 + s2].x, pf[2 * k2 + s2].y, pf[2 * k2 + s2].z, pf[2 * k2 + s2].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) l_valu += __uint_as_float(pw[q] << 16) + __uint_as_float(pw[q] & 0xffff0000u);
                    }
                }
            }
        };
        auto rescale_o = [&](auto online_tag) {
            if constexpr (decltype(online_tag)::value) {
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha_next;
            }
        };
        auto core = [&](auto online_tag) {
            f32x16 sc[2];
            uint4 pf[4];
#pragma unroll
            for (int c = 0; c < NST; ++c) {
                st_chunk(c, sc);
                soft_chunk(sc, pf, online_tag);
                rescale_o(online_tag);
                pv_chunk(c, pf);
            }
        };
        if (!(a.dbg & 1)) {
            // (a start-up stagger of waves 4-7 against their SIMD partners was measured in round 3: +-0, removed in round 6)
            if (online) core(std::true_type{}); else core(std::false_type{});
        }
        float l = (a.dbg & 1) ? 1.f : o[DB - 1][12];
        if constexpr (!G::ONES) l = (a.dbg & 1) ? 1.f : l_valu + __shfl_xor(l_valu, 32, 64);  // this lane's keys + the other half's
        const float rl = 1.0f / l;

        // =========================================================== output tile -> HBM through wave-private slabs
        __builtin_amdgcn_s_barrier();  // every wave is done with K and V: the K tile becomes the output staging area
        if constexpr (G::DEFER) {  // (in front of the output stores: the k-loop's first wait counts those stores as the youngest)
            if (has_next && wv == 0) {
#pragma unroll
                for (int p = 0; p < 4; ++p) issue_piece(lds0 + OFF_S0, 0u, p);
            }
        }
        {
            const int wy = w / nwx, wx = w - wy * nwx;
            bf16_t* obase = a.out + (int64_t)b * ntok * a.ldo + h * HD;
#pragma unroll
            for (int rnd = 0; rnd < ORND; ++rnd) {
                if (c32 / OROWS == rnd) {
#pragma unroll
                    for (int db = 0; db < DB; ++db)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int d = db * 32 + g * 8 + hh * 4;
                            if (d < HD)
                                *reinterpret_cast<uint2*>(oslab + (c32 & (OROWS - 1)) * ROW + d * 2) =
                                    make_uint2(pack_bf16(o[db][4 * g] * rl, o[db][4 * g + 1] * rl),
                                               pack_bf16(o[db][4 * g + 2] * rl, o[db][4 * g + 3] * rl));
                        }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < (OROWS * CPR + 63) / 64; ++t) {
                    const int c = el + 64 * t;
                    if (c < OROWS * CPR) {
                        const int row = c / CPR, cc = c - row * CPR;
                        const uint4 v = *reinterpret_cast<const uint4*>(oslab + row * ROW + cc * 16);
                        const int tok = win_token(wy, wx, wv * 32 + rnd * OROWS + row, a.gh, a.gw, a.sh, a.sw);
                        *reinterpret_cast<uint4*>(obase + (int64_t)tok * a.ldo + cc * 8) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        have_prev = true;
        b = nb; w = nwn; h = nh;
    }
    wait_vm<0>();
}

}  // namespace

extern "C" int swiftk_qkv_attention_fused(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* scale, void* out,
                                          int64_t ldo, int64_t K, int B, int gh, int gw, int heads, int head_dim, int shift_h,
                                          int shift_w, void* stream) {
    if (!x || !w || !scale || !out || B <= 0 || heads <= 0) return SWIFTK_EINVAL;
    if (head_dim != 80 && head_dim != 88 && head_dim != 96) return SWIFTK_ESHAPE;
    if (gh <= 0 || gw <= 0 || gh % 16 || gw % 16) return SWIFTK_ESHAPE;
    if (shift_h < 0 || shift_w < 0 || shift_h >= gh || shift_w >= gw) return SWIFTK_ESHAPE;
    // K: whole 64-element k-tiles, or ending half-way into the last one with both operands' rows extending (zero / finite
    // padded) to its end -- the contract of swiftk_gemm
    int khalf = 0;
    if (K % 64 == 32 && ldx >= K + 32 && ldw >= K + 32) {
        khalf = 1;
        K += 32;
    }
    if (K <= 0 || K % 64 || ldx < K || ldw < K || ldo < (int64_t)heads * head_dim) return SWIFTK_ESHAPE;
    if (K / 64 < 2) return SWIFTK_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15) || (ldx * 2) % 16 || (ldw * 2) % 16 || (ldo * 2) % 16)
        return SWIFTK_EALIGN;
    if ((int64_t)gh * gw * ldx * 2 >= (1ll << 32) || (int64_t)3 * head_dim * ldw * 2 >= (1ll << 32)) return SWIFTK_ESHAPE;
    FusedArgs a;
    a.x = static_cast<const char*>(x);
    a.w = static_cast<const char*>(w);
    a.out = static_cast<bf16_t*>(out);
    a.scale = scale;
    a.ldx_b = ldx * 2;
    a.ldw_b = ldw * 2;
    a.ldo = ldo;
    a.B = B; a.gh = gh; a.gw = gw; a.heads = heads; a.sh = shift_h; a.sw = shift_w;
    a.nk = (int)(K / 64);
    a.khalf = khalf;
    a.dbg = g_attn_dbg >> 8;  // tuning key 4, bits 8..: timing experiments of this kernel
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nitems = B * (gh / 16) * (gw / 16) * heads;
    int grid = g_persist_wgs >= 8 ? (g_persist_wgs & ~7) : g_persist_wgs;  // tuning key 2: one workgroup per CU the stream may use
    if (nitems < grid) grid = nitems >= 8 ? (nitems & ~7) : nitems;
    const bool timed = swiftk_prof_begin(SWIFTK_PROF_ATTENTION, 0, st);
    if (g_attn_pp) {
        if (head_dim == 80) hipLaunchKernelGGL((qkv_attn_kernel<80, true>), dim3(grid), dim3(NT), 0, st, a, nitems);
        else if (head_dim == 96) hipLaunchKernelGGL((qkv_attn_kernel<96, true>), dim3(grid), dim3(NT), 0, st, a, nitems);
        else hipLaunchKernelGGL((qkv_attn_kernel<88, true>), dim3(grid), dim3(NT), 0, st, a, nitems);
    } else {
        if (head_dim == 80) hipLaunchKernelGGL((qkv_attn_kernel<80, false>), dim3(grid), dim3(NT), 0, st, a, nitems);
        else if (head_dim == 96) hipLaunchKernelGGL((qkv_attn_kernel<96, false>), dim3(grid), dim3(NT), 0, st, a, nitems);
        else hipLaunchKernelGGL((qkv_attn_kernel<88, false>), dim3(grid), dim3(NT), 0, st, a, nitems);
    }
    if (timed) swiftk_prof_end(st);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}
