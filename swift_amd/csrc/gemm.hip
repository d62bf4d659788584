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

namespace {

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
constexpr int MI = 4, NI = 11;

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
                const float h0 = v[0] / (1.0f + expf(-v[0])) * v[1];
                const float h1 = v[2] / (1.0f + expf(-v[2])) * v[3];
                store2<OutT>(C + (int64_t)m * g.ldc + (nb >> 1), h0, h1);
            } else {
                store4<OutT>(C + (int64_t)m * g.ldc + nb, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ---- optional live timing of one GEMM flavour (bench.py's roofline leg): HIP events on the launch stream ----
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
    const bool timed = (g_prof.epilogue == EPI) && (g_prof.N == 0 || g_prof.N == g.N) && g_prof.used < Prof::MAXE;
    if (timed) {
        while (g_prof.created <= g_prof.used) {
            if (hipEventCreate(&g_prof.ev[2 * g_prof.created]) != hipSuccess) return SWIFTK_EINVAL;
            if (hipEventCreate(&g_prof.ev[2 * g_prof.created + 1]) != hipSuccess) return SWIFTK_EINVAL;
            ++g_prof.created;
        }
        (void)hipEventRecord(g_prof.ev[2 * g_prof.used], st);
    }
    hipLaunchKernelGGL(kern, dim3(ntm * g.ntn), dim3(NT), 0, st, g);
    if (timed) {
        (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], st);
        ++g_prof.used;
    }
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

template <typename T, typename OutT>
int dispatch_epi(int epi, const GemmArgs& g, hipStream_t st) {
    switch (epi) {
        case SWIFTK_EPI_NONE: return launch<T, OutT, SWIFTK_EPI_NONE>(g, st);
        case SWIFTK_EPI_BIAS_POS: return launch<T, OutT, SWIFTK_EPI_BIAS_POS>(g, st);
        case SWIFTK_EPI_SWIGLU: return launch<T, OutT, SWIFTK_EPI_SWIGLU>(g, st);
    }
    return SWIFTK_EINVAL;
}

}  // namespace

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

extern "C" int swiftk_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                           int64_t N, int64_t K, int dtype, int out_dtype, int epilogue, const float* ep0, const float* ep1,
                           int64_t pos_rows, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return SWIFTK_EINVAL;
    if (dtype != SWIFTK_F32 && dtype != SWIFTK_BF16) return SWIFTK_EINVAL;
    if (out_dtype != SWIFTK_F32 && out_dtype != dtype) return SWIFTK_EINVAL;
    const int es = dtype == SWIFTK_BF16 ? 2 : 4, os = out_dtype == SWIFTK_BF16 ? 2 : 4;
    if (K % (ROWB / es) != 0 || N % 4 != 0) return SWIFTK_ESHAPE;
    if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return SWIFTK_ESHAPE;
    if (lda < K || ldw < K) return SWIFTK_ESHAPE;
    const int ovec = (epilogue == SWIFTK_EPI_SWIGLU ? 2 : 4) * os;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || (lda * es) % 16 || (ldw * es) % 16) return SWIFTK_EALIGN;
    if (((uintptr_t)C % ovec) || (ldc * os) % ovec) return SWIFTK_EALIGN;
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
    g.ntn = (int)((N + BN - 1) / BN);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SWIFTK_BF16) {
        if (out_dtype == SWIFTK_BF16) return dispatch_epi<bf16_t, bf16_t>(epilogue, g, st);
        return dispatch_epi<bf16_t, float>(epilogue, g, st);
    }
    return dispatch_epi<float, float>(epilogue, g, st);
}
