// swiftk_gemm_modnorm_residual_pair: y = bf16(A W^T) over COMPLETE rows, ModulatedNorm + residual on the pair form of the
// stream in the same kernel -- the forecast path's wo / w2 + norm at small batch (reference swinv2.py:83-86,211-212: the
// branch's output projection, `x + ModulatedNorm(branch)`).
//
// Why a second GEMM geometry: `gemm_kernel_p`'s 256 x 352 tiles give wo / w2 (N = d = 1056) 96 tiles per unit -- one unit per
// step fills 37 % of one round of the 256 CUs, and a row's statistics span three tiles, so the norm is a separate pass over y.
// Here a workgroup owns 16 MI rows (32 or 64) x ALL d columns: 256 workgroups at one (MI = 2) or two (MI = 4) units per step,
// the row is complete in the workgroup, and y never leaves the CU.  The price is operand traffic: every workgroup streams the
// whole weight (2.2 MB for wo, 5.9 MB for w2) through L2 -> LDS, so the kernel is bound by the CU's LDS-DMA rate, not by the
// matrix pipe; at large batch the 256 x 352 tile (a quarter of the L2 -> LDS bytes per FLOP) wins and the forward keeps it.
//
// Structure: six waves, wave g owns output columns [16 NI g, 16 NI (g + 1)) (NI = 11: d = 1056; NI = 10: d = 960), i.e. the
// weight rows of that range are read by wave g ALONE.  So the weight needs no workgroup barrier: each wave streams its own
// rows through its own LDS region -- ONE k-tile (64 deep: whole 128-B lines; 32-deep steps fetched every line twice) of its
// 16 NI rows, NI blocks of 16 rows x 128 B = two 1-KiB DMA pieces each, refilled block by block: as soon as block i of k-tile
// t sits in registers its slot takes block i of k-tile t + 1, which is needed one k-tile (NI - 1 blocks of MFMAs) later.
// Counted s_waitcnt vmcnt -- VMEM retires in issue order -- keeps 2 (NI - 1) pieces per wave in flight at every read.
// Only the activation tile (16 MI rows, shared by all waves) is exchanged: 8-KiB super-stages of 4 / MI k-tiles,
// double-buffered, one s_barrier per super-stage.  A piece is 8 rows x 128 B, lane l -> row l >> 3, physical chunk l & 7
// holding logical chunk (l & 7) ^ ((row of the 16-row block >> 1) & 7); the fragment reads apply the same XOR (conflict-free
// ds_read_b128).
// Epilogue: accumulators -> bf16 (the rounding a plain bf16 GEMM applies to y) -> a row-major LDS tile over the weight ring;
// then a wave takes whole rows: exact two-pass variance, x += n P + Q on (hi + lo), new (bf16 hi, 8-bit lo) -- the arithmetic
// of modnorm_pair_kernel (elementwise.hip), next row's hi / lo loads in flight.
#include "common.h"

int g_fwd_rownorm = 0;  // tuning key 23: the forward uses this kernel for wo / w2 + norm up to this many units per step (0 = never)

int g_rownorm_dbg = 0;  // tuning key 24 (timing experiments; results are wrong while set): 1 = no row phase, 2 = one k-tile, 4 = no P / Q, 8 = no prefetch use

namespace {

constexpr int NW = 6;
constexpr int NTH = NW * 64;

struct RowNormArgs {
    const char* a;
    const char* w;
    bf16_t* xh;
    uint8_t* xl;
    const float *gamma, *beta, *mod;
    int64_t lda_b, ldw_b, ldh, ldl, ldmod, rps;
    int nk;     // k-tiles of 64
    int khalf;  // the last k-tile holds 32 valid columns
    int dbg;
    float eps;
};

template <int NI>
struct Geo {
    static constexpr int D = NW * 16 * NI;
    static constexpr int NC = D / 8;          // 16-B chunks of bf16 per row
    static constexpr int WRING = D * 128;     // one 64-deep k-tile of the weight (all six waves' regions)
    static constexpr int A_BYTES = 8192;      // one activation super-stage: 8 pieces
    static constexpr int OFF_A = WRING;
    static constexpr int OFF_PQ = OFF_A + 2 * A_BYTES;
    static constexpr int LDS_TOTAL = OFF_PQ + 2 * D * 4;
    static constexpr int YROW = D * 2 + 16;   // bytes per row of the bf16 y tile (16 B of padding: rows start 4 banks apart)
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NI, int MI>
__global__ __launch_bounds__(NTH) void gemm_rownorm_kernel(RowNormArgs a) {
    using G = Geo<NI>;
    constexpr int D = G::D, NC = G::NC, ROWS = 16 * MI;
    static_assert(ROWS * G::YROW <= G::OFF_PQ, "the y tile must fit below the P / Q vectors");
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    float* sP = reinterpret_cast<float*>(smem + G::OFF_PQ);
    float* sQ = sP + D;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    auto pin = [](const char* base) {
        const uint64_t u = (uint64_t)base;
        return (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(u >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)u));
    };

    // ---- DMA sources
    const int prow = lane >> 3, pchunk = lane & 7;
    uint32_t vW[2];  // the two pieces (rows 0-7, 8-15) of a 16-row weight block
#pragma unroll
    for (int p = 0; p < 2; ++p)
        vW[p] = (uint32_t)((16 * NI * g + 8 * p + prow) * (int)a.ldw_b) + 16u * (pchunk ^ ((4 * p + (prow >> 1)) & 7));
    // activation super-stage: 8 pieces, piece q = k-tile q / (2 MI) of the super-stage, rows 8 (q % (2 MI)) ..; this wave's
    // pieces are q = g and g + 6 (< 8)
    constexpr int AKT = 4 / MI;
    const int nA = g < 2 ? 2 : 1;
    uint32_t vA[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = g + 6 * p, rr = 8 * (q % (2 * MI)) + prow;
        vA[p] = (uint32_t)(rr * (int)a.lda_b) + 16u * (pchunk ^ (((rr & 15) >> 1) & 7));
    }
    const char* abase = pin(a.a + row0 * a.lda_b);
    const char* wbase = pin(a.w);
    const int nk = a.nk, nsup = (nk + AKT - 1) / AKT;
    const uint32_t wring = lds0 + g * (NI * 2048);
    auto issue_wblk = [&](int t, int i) {
        const char* wk = wbase + (int64_t)t * 128 + (int64_t)i * 16 * a.ldw_b;
        dma_piece_fast(wring + i * 2048, wk, vW[0]);
        dma_piece_fast(wring + i * 2048 + 1024, wk, vW[1]);
    };
    auto issue_a = [&](int j) {
        const uint32_t dst = lds0 + G::OFF_A + (j & 1) * G::A_BYTES;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int q = g + 6 * p;
            if (q < 8) {
                int kt = j * AKT + q / (2 * MI);
                kt = kt < nk ? kt : nk - 1;  // (a partial last super-stage: the piece lands, nobody reads it)
                dma_piece_fast(dst + q * 1024, abase + (int64_t)kt * 128, vA[p]);
            }
        }
    };

    // ---- k-loop
    f32x4 acc[MI][NI];
#pragma unroll
    for (int r = 0; r < MI; ++r)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment reads: lane l -> row l & 15 of a 16-row block, logical chunk (l >> 4) + 4 (k-half)
    const int fr = lane & 15;
    const uint32_t rd0 = (uint32_t)(fr * 128 + 16 * ((lane >> 4) ^ ((fr >> 1) & 7)));
    const uint32_t rd1 = (uint32_t)(fr * 128 + 16 * (((lane >> 4) + 4) ^ ((fr >> 1) & 7)));
    issue_a(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_wblk(0, i);
    // (the first k-tile is on its way while the modulation vectors are folded)
    // ---- P = gamma (1 + scale), Q = beta (1 + scale) + shift of this workgroup's sample (its rows are of one sample)
    {
        const float* mrow = a.mod + (row0 / a.rps) * a.ldmod;
        for (int c = tid; c < NC; c += NTH) {
            float gm[8], bt[8], sc[8], sh[8], p[8], q[8];
            load8<float>(a.gamma + 8 * c, gm);
            load8<float>(a.beta + 8 * c, bt);
            load8<float>(mrow + 8 * c, sc);
            load8<float>(mrow + D + 8 * c, sh);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p[e] = gm[e] * (1.0f + sc[e]);
                q[e] = bt[e] * (1.0f + sc[e]) + sh[e];
            }
            store8<float>(sP + 8 * c, p);
            store8<float>(sQ + 8 * c, q);
        }
    }
    // the rows this wave finishes in the epilogue are r = g + 6 k; the residual stream's (hi, lo) of the first PF of them are
    // requested now and arrive under the k-loop (the compiler's loads share the DMA's counter and are OLDER than every piece
    // issued from here on, so the counted waits below can only wait longer, never too short)
    struct Row {
        uint4 h[3];
        uint2 l[3];
    };
    auto load_row = [&](int r, Row& rw) {
        const bf16_t* hp = a.xh + (row0 + r) * a.ldh;
        const uint8_t* lp = a.xl + (row0 + r) * a.ldl;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int c = lane + 64 * s;
            if (c < NC) {
                rw.h[s] = *reinterpret_cast<const uint4*>(hp + 8 * c);
                rw.l[s] = *reinterpret_cast<const uint2*>(lp + 8 * c);
            }
        }
    };
    constexpr int KROWS = (ROWS + NW - 1) / NW;  // rows per wave (the last one only for the first ROWS % 6 waves)
    constexpr int BR = 6;                        // rows per epilogue batch
    constexpr int PF = MI == 2 ? BR : 0;         // (MI = 4 has no registers to spare under the k-loop)
    Row rows[BR];
#pragma unroll
    for (int k = 0; k < PF; ++k)
        if (g + NW * k < ROWS) load_row(g + NW * k, rows[k]);
    wait_vm<0>();
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int j = t / AKT, kk = t - j * AKT;
        const bool a_now = kk == 0 && j + 1 < nsup, last = t + 1 == nk;
        if (a_now) issue_a(j + 1);
        const char* ap = smem + G::OFF_A + (j & 1) * G::A_BYTES + kk * (MI * 2048);
        uint4 xf[MI][2];
#pragma unroll
        for (int r = 0; r < MI; ++r) {
            xf[r][0] = *reinterpret_cast<const uint4*>(ap + r * 2048 + rd0);
            xf[r][1] = *reinterpret_cast<const uint4*>(ap + r * 2048 + rd1);
        }
        const bool half = last && a.khalf;
        if (last) wait_vm<0>();
        const char* wp = smem + g * (NI * 2048);
        // block i of this k-tile has landed when everything issued after it may still be in flight: blocks i + 1 .. of this
        // k-tile, this k-tile's activation request, blocks .. i - 1 of the next k-tile
        auto wait_blk = [&]() {
            if (!last) {
                if (!a_now) wait_vm<2 * (NI - 1)>();
                else if (nA == 2) wait_vm<2 * (NI - 1) + 2>();
                else wait_vm<2 * (NI - 1) + 1>();
            }
        };
        // software pipeline over the blocks: block i + 1 is read while block i's MFMAs issue; once it sits in registers its
        // slot takes the same block of the next k-tile
        uint4 wf[2][2];
        wait_blk();
        wf[0][0] = *reinterpret_cast<const uint4*>(wp + rd0);
        wf[0][1] = *reinterpret_cast<const uint4*>(wp + rd1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!last) issue_wblk(t + 1, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) {
                wait_blk();
                wf[(i + 1) & 1][0] = *reinterpret_cast<const uint4*>(wp + (i + 1) * 2048 + rd0);
                wf[(i + 1) & 1][1] = *reinterpret_cast<const uint4*>(wp + (i + 1) * 2048 + rd1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < MI; ++r)
                acc[r][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i & 1][0]),
                                                                    __builtin_bit_cast(bf16x8, xf[r][0]), acc[r][i], 0, 0, 0);
            if (!half) {
#pragma unroll
                for (int r = 0; r < MI; ++r)
                    acc[r][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i & 1][1]),
                                                                        __builtin_bit_cast(bf16x8, xf[r][1]), acc[r][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < NI) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!last) issue_wblk(t + 1, i + 1);
            }
        }
        // super-stage hand-over: everybody is done with this activation buffer, and the next one -- requested at the top of this
        // super-stage, older than the 2 NI weight pieces issued since -- is complete
        if (kk == AKT - 1 && !last) {
            wait_vm<2 * NI>();
            __syncthreads();
        }
    }

    // ---- epilogue.  Lane l of an accumulator block holds row l & 15, columns 4 (l >> 4) .. + 3 (MFMA A := weight rows).
    __syncthreads();  // every wave has left the k-loop: the rings are free
#pragma unroll
    for (int r = 0; r < MI; ++r)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = 16 * r + (lane & 15), c0 = 16 * NI * g + 16 * i + 4 * (lane >> 4);
            *reinterpret_cast<uint2*>(smem + m * G::YROW + 2 * c0) =
                make_uint2(pack_bf16(acc[r][i][0], acc[r][i][1]), pack_bf16(acc[r][i][2], acc[r][i][3]));
        }
    if constexpr (PF == 0) {  // (after the accumulators have left their registers)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < BR; ++k)
            if (g + NW * k < ROWS) load_row(g + NW * k, rows[k]);
    }
    __syncthreads();
    if (a.dbg & 1) return;
    // a batch of up to BR rows per wave: the three passes (sum, centred squares, update) each walk the whole batch, so the
    // batch's wave reductions overlap; y is re-read from LDS and unpacked in every pass instead of being kept
    auto y_chunk = [&](int r, int c, float (&v)[8]) { load8<bf16_t>(reinterpret_cast<const bf16_t*>(smem + r * G::YROW) + 8 * c, v); };
#pragma unroll
    for (int k0 = 0; k0 < KROWS; k0 += BR) {
        if (k0 > 0) {
#pragma unroll
            for (int k = 0; k < BR; ++k)
                if (k0 + k < KROWS && g + NW * (k0 + k) < ROWS) load_row(g + NW * (k0 + k), rows[k]);
        }
        float mean[BR], rstd[BR];
#pragma unroll
        for (int k = 0; k < BR; ++k) {
            const int r = g + NW * (k0 + k);
            float sum = 0.f;
            if (k0 + k < KROWS && r < ROWS) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int c = lane + 64 * s;
                    if (c < NC) {
                        float v[8];
                        y_chunk(r, c, v);
                        sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                    }
                }
            }
            mean[k] = sum;
        }
#pragma unroll
        for (int k = 0; k < BR; ++k) mean[k] = wave_sum(mean[k]) / (float)D;
#pragma unroll
        for (int k = 0; k < BR; ++k) {
            const int r = g + NW * (k0 + k);
            float sq = 0.f;
            if (k0 + k < KROWS && r < ROWS) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int c = lane + 64 * s;
                    if (c < NC) {
                        float v[8];
                        y_chunk(r, c, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v[e] -= mean[k];
                            sq += v[e] * v[e];
                        }
                    }
                }
            }
            rstd[k] = sq;
        }
#pragma unroll
        for (int k = 0; k < BR; ++k) rstd[k] = rsqrtf(wave_sum(rstd[k]) / (float)D + a.eps);
#pragma unroll
        for (int k = 0; k < BR; ++k) {
            const int r = g + NW * (k0 + k);
            if (!(k0 + k < KROWS && r < ROWS)) continue;
            const Row& rw = rows[k];
            bf16_t* hp = a.xh + (row0 + r) * a.ldh;
            uint8_t* lp = a.xl + (row0 + r) * a.ldl;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int c = lane + 64 * s;
                if (c < NC) {
                    float v[8], P[8], Q[8];
                    y_chunk(r, c, v);
                    load8<float>(sP + 8 * c, P);
                    load8<float>(sQ + 8 * c, Q);
                    const uint32_t hw[4] = {rw.h[s].x, rw.h[s].y, rw.h[s].z, rw.h[s].w};
                    const uint32_t lw[2] = {rw.l[s].x, rw.l[s].y};
                    float xn[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float hi = (e & 1) ? __uint_as_float(hw[e >> 1] & 0xffff0000u) : __uint_as_float(hw[e >> 1] << 16);
                        const uint32_t E = (hw[e >> 1] >> ((e & 1) ? 23 : 7)) & 0xFFu;
                        const float b = (float)((lw[e >> 2] >> (8 * (e & 3))) & 0xFFu);
                        xn[e] = (hi + lo8_value(b, E)) + (((v[e] - mean[k]) * rstd[k]) * P[e] + Q[e]);
                    }
                    uint32_t oh[4], ol[2] = {0u, 0u};
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const uint32_t ph = pack_bf16(xn[2 * e2], xn[2 * e2 + 1]);
                        oh[e2] = ph;
                        ol[e2 >> 1] = lo8_insert(xn[2 * e2], __uint_as_float(ph << 16), (ph >> 7) & 0xFFu, (2 * e2) & 3, ol[e2 >> 1]);
                        ol[e2 >> 1] = lo8_insert(xn[2 * e2 + 1], __uint_as_float(ph & 0xffff0000u), (ph >> 23) & 0xFFu, (2 * e2 + 1) & 3, ol[e2 >> 1]);
                    }
                    *reinterpret_cast<uint4*>(hp + 8 * c) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
                    *reinterpret_cast<uint2*>(lp + 8 * c) = make_uint2(ol[0], ol[1]);
                }
            }
        }
    }
}

}  // namespace

extern "C" int swiftk_gemm_modnorm_residual_pair(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t K, void* x_hi,
                                                 int64_t ldh, void* x_lo, int64_t ldl, const float* gamma, const float* beta,
                                                 const float* mod, int64_t ldmod, int64_t M, int d, int64_t rows_per_sample,
                                                 float eps, int rows_per_workgroup, void* stream) {
    if (!A || !W || !x_hi || !x_lo || !gamma || !beta || !mod || M <= 0) return SWIFTK_EINVAL;
    if (d != 1056 && d != 960) return SWIFTK_ESHAPE;
    if (rows_per_workgroup != 32 && rows_per_workgroup != 64) return SWIFTK_ESHAPE;
    // (whole 128-B lines are fetched: a K that ends half-way into its last 64-wide k-tile needs the rows padded to the tile)
    if (K < 64 || K % 32 || lda < (K + 63) / 64 * 64 || ldw < (K + 63) / 64 * 64 || ldh < d || ldl < d) return SWIFTK_ESHAPE;
    if (M % rows_per_workgroup || rows_per_sample <= 0 || rows_per_sample % rows_per_workgroup || M % rows_per_sample)
        return SWIFTK_ESHAPE;
    if (M / rows_per_workgroup > 0x7fffffff || (int64_t)d * ldw * 2 > 0x7fffffffLL || 64 * lda * 2 > 0x7fffffffLL) return SWIFTK_ESHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)x_hi & 15) || ((uintptr_t)x_lo & 7) || ((uintptr_t)gamma & 15) ||
        ((uintptr_t)beta & 15) || ((uintptr_t)mod & 15) || (lda * 2) % 16 || (ldw * 2) % 16 || (ldh * 2) % 16 || ldl % 8 || ldmod % 4)
        return SWIFTK_EALIGN;
    RowNormArgs a;
    a.a = static_cast<const char*>(A);
    a.w = static_cast<const char*>(W);
    a.xh = static_cast<bf16_t*>(x_hi);
    a.xl = static_cast<uint8_t*>(x_lo);
    a.gamma = gamma; a.beta = beta; a.mod = mod;
    a.lda_b = lda * 2; a.ldw_b = ldw * 2; a.ldh = ldh; a.ldl = ldl; a.ldmod = ldmod; a.rps = rows_per_sample;
    a.nk = (int)((K + 63) / 64);
    a.khalf = (K % 64) != 0;
    a.dbg = g_rownorm_dbg;
    if (a.dbg & 2) a.nk = 1;
    a.eps = eps;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = (int)(M / rows_per_workgroup);
    if (d == 1056) {
        if (rows_per_workgroup == 32) hipLaunchKernelGGL((gemm_rownorm_kernel<11, 2>), dim3(grid), dim3(NTH), 0, st, a);
        else hipLaunchKernelGGL((gemm_rownorm_kernel<11, 4>), dim3(grid), dim3(NTH), 0, st, a);
    } else {
        if (rows_per_workgroup == 32) hipLaunchKernelGGL((gemm_rownorm_kernel<10, 2>), dim3(grid), dim3(NTH), 0, st, a);
        else hipLaunchKernelGGL((gemm_rownorm_kernel<10, 4>), dim3(grid), dim3(NTH), 0, st, a);
    }
    SWIFTK_CHECK_LAUNCH();
    return 0;
}
