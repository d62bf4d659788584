// Weight-gradient GEMM in TN form for gfx950:  C[N1, N2] = sum_m P[m, N1] * Q[m, N2]  (bf16 operands, fp32 slabs).
//
// The weight gradients of a Linear (reference: autograd of `swinv2.py:112-113,134,96-98`) contract over the TOKEN index, and
// both operands (dY [tokens, out], X [tokens, in]) are stored token-major.  Feeding them to the NT kernel of gemm.hip needs
// two transposed copies per gradient (7.5 % of a CRPS iteration).  Here the operands stay as they are:
//   * a k-tile is 64 token rows; LDS-DMA (`global_load_lds_dwordx4`) lays it down as 8-row x 32-column subtiles of 512 B
//     (cdna_hip_programming.md T10, image (a)): chunk c of row r of a subtile sits at 64*(r&7) + 16*(c ^ ((r>>2)&3)), the
//     XOR applied to the per-lane SOURCE address since the DMA writes lane-linear;
//   * MFMA operands are read with `ds_read_b64_tr_b16`: per 16-lane group a 4-row x 16-column block delivered column-major,
//     so lane (n = l&15, g = l>>4) receives k = 8g + 4h + {0..3} of column n -- the same (lane, element) -> k map as the row
//     reads of gemm.hip, hence the same products summed in the same order: results are bit-equal to the NT kernel run on
//     transposed copies.  The two blocks of a 32-lane half sit 8 rows apart in the same columns: conflict-free (T10).
// Everything else (256 x 320 / 352 / 384 tiles, 8 waves 4 x 2, two 72-80-KiB stages, DMA pieces issued between MFMA groups,
// persistent grid over (tile, k-split) items, XCD-grouped tile order, fp32 slab store) follows gemm_kernel_p.
#include "common.h"

#ifndef SWIFTK_TN_G0_FIRST
#define SWIFTK_TN_G0_FIRST 5  // pieces waves 0-3 issue in a k-tile's first phase (the rest in the second)
#endif

namespace {

constexpr int BM = 256, KT = 64;
constexpr int A_RG = BM / 32 * 512;        // bytes of one 8-row group of the P image: 8 subtiles
constexpr int A_BYTES = 8 * A_RG;          // 32 KiB
constexpr int NT = 512;
constexpr int MI = 4;
// NI = W-side MFMA tiles per wave: 10 / 11 / 12 -> 320 / 352 / 384 output columns per tile (as gemm_kernel_p): the Q image
// has NI subtiles per 8-row group, a stage is 72 / 76 / 80 KiB

struct TnArgs {
    const char* P;
    const char* Q;
    float* C;
    int64_t ldp_b, ldq_b;  // row strides in bytes
    int64_t ldc, c_split;  // fp32 elements
    int N1, N2, K;         // K = contracted rows (tokens), a multiple of 64
    int ntn, ksplit;
};

struct TileIter {
    int ntm, ntn, gm;
    __device__ __forceinline__ void coords(int t, int& tm, int& tn) const {
        const int per = gm * ntn;
        const int grp = t / per, r = t - grp * per;
        const int rows = min(gm, ntm - grp * gm);
        tn = r / rows;
        tm = grp * gm + (r - tn * rows);
    }
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 tr_pair(const char* lo, const char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi);
    return __builtin_bit_cast(uint4, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// PP: ping-pong k-loop (the schedule of gemm.hip's persistent kernel, section "The k-loop" of DESIGN.md): a k-tile is 2 NP phases
// (32-row k-step x a third / a half of the wave tile's column blocks); waves 4-7 run one barrier behind waves 0-3.  Here a k-tile's
// rows split by WAVE: row group w (k rows 8 w .. 8 w + 7) is filled by wave w, so the first k-step reads what waves 0-3 requested
// and the second what waves 4-7 requested -- waves 0-3 issue their pieces in the first phases and wait for them in the k-tile's
// last MEM phase, waves 4-7 spread theirs over all phases, leave them in flight across the k-tile boundary and retire them
// (counted) in front of the next k-tile's second k-step.
template <int NI, bool PP>
__global__ __launch_bounds__(NT) void gemm_tn_kernel(TnArgs g, int ntm, int gm) {
    constexpr int WT = 16 * NI, BN = 2 * WT;
    constexpr int B_RG = NI * 512, B_BYTES = 8 * B_RG, STAGE = A_BYTES + B_BYTES;
    constexpr bool ODD = NI & 1;  // 11 subtiles per row group do not pair up: one piece straddles two row groups
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const TileIter it{ntm, g.ntn, gm};
    const int ksplit = g.ksplit;
    const int ntiles = ntm * g.ntn * ksplit;
    int vid;
    {  // workgroups with equal blockIdx%8 share an XCD: give each XCD a contiguous run of work items
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int stride = gridDim.x;
    if (vid >= ntiles) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

    // ---- DMA sources.  A 1-KiB piece = two neighbouring subtiles.  Wave w fills row group w of the P image (4 pieces) and of
    // the Q image: with an even subtile count (NI = 10 / 12) the NI / 2 pairs of row group w; with 11 subtiles per row group either the pairs (0,1)..(8,9) of row group 2G plus the piece made of
    // (2G, 10) and (2G+1, 0) [even w], or the pairs (1,2)..(9,10) of row group 2G+1 [odd w], G = w>>1.  Lane l writes LDS
    // byte 16*l of its piece: subtile l>>5, row (l&31)>>2, position l&3 <- chunk (l&3) ^ ((row>>2)&3).
    const int dhalf = lane >> 5, drow = (lane & 31) >> 2, dslot = lane & 3, dhb = (lane >> 4) & 1;
    const uint32_t va = (uint32_t)(drow * g.ldp_b) + 64u * dhalf + 16u * (dslot ^ (2 * (wv & 1) | dhb));
    const uint32_t vb = (uint32_t)(drow * g.ldq_b) + 64u * dhalf + 16u * (dslot ^ (2 * (wv & 1) | dhb));
    const uint32_t vs = dhalf ? (uint32_t)((8 + drow) * g.ldq_b) - 640u + 16u * (dslot ^ (2 | dhb))
                              : (uint32_t)(drow * g.ldq_b) + 16u * (dslot ^ dhb);  // (NI = 11 only)
    const int nk_all = g.K / KT;
    auto k_begin = [&](int item) { return (int)((int64_t)(item % ksplit) * nk_all / ksplit); };
    auto k_end = [&](int item) { return (int)((int64_t)(item % ksplit + 1) * nk_all / ksplit); };
    const char* abase[4];
    const char* qbase[6];
    // piece bases of work item `t`, at the first k-tile of its k-range (the per-lane offset then advances by whole k-tiles)
    auto set_sources = [&](int t) {
        int tm, tn;
        it.coords(t / ksplit, tm, tn);
        const int64_t k0 = (int64_t)k_begin(t) * KT;
        const int n1 = tm * BM, n2 = tn * BN;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int c = n1 + 64 * p;
            c = c < g.N1 ? c : n1;  // columns past the matrix feed accumulators that are never stored: any readable address
            abase[p] = g.P + (k0 + 8 * wv) * g.ldp_b + 2 * c;
        }
        if constexpr (ODD) {
            const int G = wv >> 1;
#pragma unroll
            for (int y = 0; y < 5; ++y)
                qbase[y] = g.Q + (k0 + 16 * G + 8 * (wv & 1)) * g.ldq_b + 2 * (n2 + 32 * (wv & 1) + 64 * y);
            qbase[5] = g.Q + (k0 + 16 * G) * g.ldq_b + 2 * (n2 + 320);
        } else {  // NI / 2 aligned pairs of row group wv; columns past the matrix are redirected like P's
#pragma unroll
            for (int y = 0; y < NI / 2; ++y) {
                int c = n2 + 64 * y;
                c = c < g.N2 ? c : n2;
                qbase[y] = g.Q + (k0 + 8 * wv) * g.ldq_b + 2 * c;
            }
        }
    };
    auto issue_piece = [&](uint32_t sa, uint32_t koff_a, uint32_t koff_q, int p) {
        if (p < 4) {
            dma_piece_fast(sa + wv * A_RG + p * 1024, abase[p], va + koff_a);
        } else if constexpr (ODD) {
            if (p < 9) dma_piece_fast(sa + A_BYTES + wv * B_RG + 512 * (wv & 1) + (p - 4) * 1024, qbase[p - 4], vb + koff_q);
            else if (!(wv & 1)) dma_piece_fast(sa + A_BYTES + wv * B_RG + 5120, qbase[5], vs + koff_q);
        } else {
            if (p - 4 < NI / 2) dma_piece_fast(sa + A_BYTES + wv * B_RG + (p - 4) * 1024, qbase[p - 4], vb + koff_q);
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses: lane = (g = l>>4: k rows 8g..8g+7 of a 32-row step, q = (l>>2)&3, p = l&3); read h (0/1) takes
    // rows 8g + 4h + q, columns c0 + 4p..4p+3 of the 16-column block c0 = 16 * tile index; the block's chunk inside its
    // subtile is 2*((c0>>4)&1) + (p>>1).  Tile-index parity and h change the XOR per lane, so each gets its own register;
    // subtile column, 32-row step and stage are immediates / scalar adds.
    const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
    int vA[2][2], vW[2][2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int xr = 2 * (fg & 1) | h;
            vA[par][h] = fg * A_RG + 512 * (2 * wm) + 64 * (4 * h + fq) + 16 * ((2 * par + (fp >> 1)) ^ xr) + 8 * (fp & 1);
            // block c0 = wn * WT + 16 j: subtile (c0 >> 5) = wn * (WT / 32) + (j >> 1) [+ (j & 1) when WT / 16 is odd and wn = 1],
            // chunk bit (c0 >> 4) & 1 = (j & 1) ^ (wn & ODD)
            vW[par][h] = A_BYTES + fg * B_RG + 512 * (wn ? (WT / 32) + (ODD ? par : 0) : 0) + 64 * (4 * h + fq) +
                         16 * ((2 * (par ^ (ODD ? wn : 0)) + (fp >> 1)) ^ xr) + 8 * (fp & 1);
        }

    int tile = vid, kt = 0;
    set_sources(tile);
    int nk = k_end(tile) - k_begin(tile);
#pragma unroll
    for (int p = 0; p < 10; ++p) issue_piece(lds0, 0u, 0u, p);
    int par = 0;
    if (!PP && wv >= 4) __builtin_amdgcn_s_setprio(1);
    const uint32_t kstep_a = (uint32_t)(KT * g.ldp_b), kstep_q = (uint32_t)(KT * g.ldq_b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int grp = wv >> 2;
    bool first_kt = true;
    for (;;) {
        if constexpr (PP) {
            if (first_kt) {  // (inside an item the phase barriers hand the stages over)
                __builtin_amdgcn_s_barrier();
                if (grp) __builtin_amdgcn_s_barrier();
            }
        } else {
            __builtin_amdgcn_s_barrier();
        }
        const char* s = smem + par * STAGE;
        const uint32_t fill = lds0 + (par ^ 1) * STAGE;
        const bool last_k = (kt + 1 == nk);
        uint32_t koff_a = (uint32_t)(kt + 1) * kstep_a, koff_q = (uint32_t)(kt + 1) * kstep_q;
        if (last_k) {
            const int ntile = tile + stride;
            set_sources(ntile < ntiles ? ntile : tile);
            koff_a = koff_q = 0;
        }
        if constexpr (PP) {
            constexpr int NP = NI == 12 ? 3 : 2, PS = (NI + NP - 1) / NP, NQ = 2 * NP;
            uint4 xf[MI], wf[PS];
            auto bar = [] {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ks = q / NP, j0 = (q % NP) * PS, nj = NI - j0 < PS ? NI - j0 : PS;
                if (q % NP == 0) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int o = 512 * (i >> 1) + ks * 4 * A_RG;
                        xf[i] = tr_pair(s + vA[i & 1][0] + o, s + vA[i & 1][1] + o);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < PS; ++jj) {
                    if (jj < nj) {
                        const int j = j0 + jj, o = 512 * (j >> 1) + ks * 4 * B_RG;
                        wf[jj] = tr_pair(s + vW[j & 1][0] + o, s + vW[j & 1][1] + o);
                    }
                }
                if (grp) {  // waves 4-7: ten pieces over all phases; the previous k-tile's pieces retire behind phase NP - 1
#pragma unroll
                    for (int pc = 0; pc < 10; ++pc)
                        if (pc * NQ / 10 == q) issue_piece(fill, koff_a, koff_q, pc);
                    // (pieces 0..4 -- the four P pieces and the first Q piece, real for every wave -- are this k-tile's requests so far)
                    static_assert(4 * NQ / 10 == NP - 1 && 5 * NQ / 10 == NP, "phases 0 .. NP - 1 of waves 4-7 carry pieces 0 .. 4");
                    if (q == NP - 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                } else {    // waves 0-3: everything in the first two phases, waited for in the last one.  (Their rows are the next
                            // k-tile's FIRST k-step: spread over NQ - 1 phases the last pieces had one phase to land and the wait
                            // stalled every k-tile -- 4, 5 or 6 pieces up front measure alike, 7 and all ten cost again)
#pragma unroll
                    for (int pc = 0; pc < 10; ++pc)
                        if ((pc < SWIFTK_TN_G0_FIRST ? 0 : 1) == q) issue_piece(fill, koff_a, koff_q, pc);
                    if (q == NQ - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (q == NQ - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bar();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int jj = 0; jj < PS; ++jj) {
                    if (jj < nj) {
#pragma unroll
                        for (int i = 0; i < MI; ++i)
                            acc[i][j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[jj]),
                                                                                     __builtin_bit_cast(bf16x8, xf[i]), acc[i][j0 + jj], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                bar();
            }
        } else {
            auto k_half = [&](const int ks, const bool with_dma) {
                uint4 xf[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int o = 512 * (i >> 1) + ks * 4 * A_RG;
                    xf[i] = tr_pair(s + vA[i & 1][0] + o, s + vA[i & 1][1] + o);
                }
                auto wfrag = [&](int j) {
                    const int o = 512 * (j >> 1) + ks * 4 * B_RG;
                    return tr_pair(s + vW[j & 1][0] + o, s + vW[j & 1][1] + o);
                };
                constexpr int AHEAD = NI == 12 ? 1 : 2;  // W fragments in flight (192 accumulators leave room for one)
                uint4 wf = wfrag(0);
                uint4 wf1 = wfrag(AHEAD == 2 ? 1 : 0);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    uint4 wn_ = wf1;
                    if constexpr (AHEAD == 2) {
                        if (j + 2 < NI) wf1 = wfrag(j + 2);
                    } else {
                        if (j + 1 < NI) wn_ = wfrag(j + 1);
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf),
                                                                            __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
                    if (with_dma && j < 10) issue_piece(fill, koff_a, koff_q, j);
                    wf = wn_;
                }
            };
            k_half(0, true);
            k_half(1, false);
        }
        par ^= 1;
        if (!last_k) {
            ++kt;
            first_kt = false;
            if constexpr (!PP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        if constexpr (PP) {
            if (!grp) __builtin_amdgcn_s_barrier();  // waves 0-3 meet waves 4-7: every wave aligned again
            first_kt = true;
        }
        // ---- store this item's slab: lane holds C[n1 = ..+r16][n2 = ..+4*(lane>>4) .. +3]
        {
            int tm, tn;
            it.coords(tile / ksplit, tm, tn);
            float* C = g.C + (int64_t)(tile % ksplit) * g.c_split;
            int elane = lane;
            asm volatile("" : "+v"(elane));
            const int r16 = elane & 15, g4 = elane >> 4;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = tm * BM + wm * 64 + i * 16 + r16;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int nb = tn * BN + wn * WT + j * 16 + 4 * g4;
                    const f32x4 v = acc[i][j];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (m < g.N1 && nb < g.N2)
                        *reinterpret_cast<float4*>(C + (int64_t)m * g.ldc + nb) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        tile += stride;
        if (tile >= ntiles) break;
        kt = 0;
        nk = k_end(tile) - k_begin(tile);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing dummy DMA must not outlive the LDS allocation
}

}  // namespace

int g_tn_pp = 1;  // tuning key 22: ping-pong k-loop of the weight-gradient GEMM: 1 always (default), 2 352-wide tiles only, 0 never (384-wide tiles: always)

// slabs[s][N1][N2] (fp32, row stride ldc, slab stride `slab_stride`) = partial products over the s-th of `ksplit` ranges of
// the K token rows.  P: [K, >= N1] bf16 with row stride ldp, Q: [K, >= N2] with ldq.  Shapes this kernel does not take
// (SWIFTK_ESHAPE: token counts that are not whole 64-row k-tiles, rows too short to read whole 128-B segments) go through
// swiftk_transpose + swiftk_gemm_splitk instead.
extern "C" int swiftk_gemm_tn_splitk(const void* P, int64_t ldp, const void* Q, int64_t ldq, float* slabs, int64_t ldc,
                                     int64_t slab_stride, int64_t N1, int64_t N2, int64_t K, int ksplit, void* stream) {
    if (!P || !Q || !slabs || N1 <= 0 || N2 <= 0 || K <= 0 || ksplit < 1) return SWIFTK_EINVAL;
    if (ksplit > 1 && slab_stride < N1 * ldc) return SWIFTK_EINVAL;
    if (K % KT || N1 % 8 || N2 % 4 || K / KT < ksplit) return SWIFTK_ESHAPE;
    // tile width: 352 where it divides N2 (Swift-B), else the even forms (320 / 384), whichever wastes fewer columns.  Whole
    // 128-B source segments: pieces are 64 columns wide at multiples of 64 -- except the 352-wide form, whose odd subtile
    // count makes it read its Q tiles in full
    int ni = 11;
    if (N2 % 352) {
        const int64_t w10 = (N2 + 319) / 320 * 320, w12 = (N2 + 383) / 384 * 384;
        ni = w12 <= w10 ? 12 : 10;
    }
    const int bn = 32 * ni;
    if (ldp < (N1 + 63) / 64 * 64 || ldq < (ni == 11 ? (N2 + bn - 1) / bn * bn : (N2 + 63) / 64 * 64) || ldc < N2) return SWIFTK_ESHAPE;
    if (N1 > (1 << 30) || N2 > (1 << 30) || K > (1 << 30)) return SWIFTK_ESHAPE;
    // the per-lane DMA offset is 32 bits and runs over one split's rows
    const int64_t rows_per_split = (K / KT + ksplit - 1) / ksplit * KT + 16;
    if (rows_per_split * ldp * 2 >= (1ll << 32) || rows_per_split * ldq * 2 >= (1ll << 32)) return SWIFTK_ESHAPE;
    if (((uintptr_t)P & 15) || ((uintptr_t)Q & 15) || (ldp * 2) % 16 || (ldq * 2) % 16 || ((uintptr_t)slabs & 15) || (ldc * 4) % 16)
        return SWIFTK_EALIGN;
    TnArgs g;
    g.P = static_cast<const char*>(P);
    g.Q = static_cast<const char*>(Q);
    g.C = slabs;
    g.ldp_b = ldp * 2;
    g.ldq_b = ldq * 2;
    g.ldc = ldc;
    g.c_split = slab_stride;
    g.N1 = (int)N1;
    g.N2 = (int)N2;
    g.K = (int)K;
    g.ntn = (int)((N2 + bn - 1) / bn);
    g.ksplit = ksplit;
    const int ntm = (int)((N1 + BM - 1) / BM);
    const int items = ntm * g.ntn * ksplit;
    const dim3 grid(items < 256 ? items : 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // (measured interleaved in one process at local batch 8, tools/tn_ab.py, with waves 0-3 issuing their pieces in a k-tile's first
    // two phases: to_qkv's gradient -6.5 %, w1's -6..-9 %, w2's -3 %, wo's 0..-5 % against one barrier per k-tile; round 6, the larger
    // variants' widths, `tn_ab.py 8 7 1280` / `1536`, bit-equal: 320-wide tiles -3.4 % per layer's gradients, 384-wide -10 %.  The
    // 384-wide one-barrier form spilled three registers and is gone: key 22 = 0 / 2 still take the ping-pong loop there)
    const bool pp = g_tn_pp == 1 || (g_tn_pp == 2 && ni == 11) || ni == 12;
    if (pp) {
        if (ni == 10) hipLaunchKernelGGL((gemm_tn_kernel<10, true>), grid, dim3(NT), 0, st, g, ntm, 8);
        else if (ni == 12) hipLaunchKernelGGL((gemm_tn_kernel<12, true>), grid, dim3(NT), 0, st, g, ntm, 8);
        else hipLaunchKernelGGL((gemm_tn_kernel<11, true>), grid, dim3(NT), 0, st, g, ntm, 8);
    } else {
        if (ni == 10) hipLaunchKernelGGL((gemm_tn_kernel<10, false>), grid, dim3(NT), 0, st, g, ntm, 8);
        else hipLaunchKernelGGL((gemm_tn_kernel<11, false>), grid, dim3(NT), 0, st, g, ntm, 8);
    }
    SWIFTK_CHECK_LAUNCH();
    return 0;
}
