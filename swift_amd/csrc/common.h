// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of swift_amd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/swiftk.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef uint16_t bf16_t;  // storage type of a bf16 element

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

#define SWIFTK_CHECK_LAUNCH()                         \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

// fp32 -> bf16, round to nearest even; a plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaN a NaN.
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// two fp32 -> one dword of two bf16 (RNE): the vector convert lowers to ONE v_cvt_pk_bf16_f32; converting the halves
// separately and or-ing them costs four VALU instructions
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T>
struct elem;
template <>
struct elem<float> {
    static constexpr int dtype = SWIFTK_F32;
    __device__ static float to_f(float v) { return v; }
    __device__ static float from_f(float v) { return v; }
};
template <>
struct elem<bf16_t> {
    static constexpr int dtype = SWIFTK_BF16;
    __device__ static float to_f(bf16_t v) { return bf2f(v); }
    __device__ static bf16_t from_f(float v) { return f2bf(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// One LDS-DMA piece, hand-issued: M0 <- LDS byte address of the piece (wave-uniform), then
// global_load_lds_dwordx4 voff, s[base:base+1].  Inline asm keeps hipcc's waitcnt pass out of the picture (it would
// otherwise drain vmcnt(0) before every ds_read that might alias the DMA target); the kernel waits for its own DMA
// with an explicit vmcnt(0) before the barrier that hands a stage over.  `s_nop 4`: SALU-written SGPRs read by VMEM.
__device__ __forceinline__ void dma_piece(uint32_t lds_dst, const char* base, uint32_t voff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds_dst), "s"(base)
        : "memory");
}

// Same, for bases that were computed well before (no SALU-write -> VMEM-read hazard to pad) and callers that do
// not need M0 preserved (nothing else in these kernels uses it): M0, one wait state, load.
__device__ __forceinline__ void dma_piece_fast(uint32_t lds_dst, const char* base, uint32_t voff) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 1\n\t"
        "global_load_lds_dwordx4 %0, %2"
        :
        : "v"(voff), "s"(lds_dst), "s"(base)
        : "memory");
}

// Same with the non-temporal cache policy: for operands that are dead in L2 once the workgroups sharing them have passed
// (the activation panel of a GEMM tile row), so that the lines of the re-read operand (the weight panel) survive them
__device__ __forceinline__ void dma_piece_fast_nt(uint32_t lds_dst, const char* base, uint32_t voff) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 1\n\t"
        "global_load_lds_dwordx4 %0, %2 nt"
        :
        : "v"(voff), "s"(lds_dst), "s"(base)
        : "memory");
}

// L2 look-ahead: a 4-byte-per-lane LDS-DMA whose only purpose is to pull each lane's 128-B line into the XCD's L2 (the
// 256 bytes it writes go to a scratch area of LDS)
__device__ __forceinline__ void dma_touch(uint32_t lds_dst, const char* base, uint32_t voff) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dword %0, %2"
        :
        : "v"(voff), "s"(lds_dst), "s"(base)
        : "memory");
}

// eight consecutive channels per lane: 16 B of bf16 / 32 B of fp32 per access
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(u[e] << 16);
        v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
    }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) =
        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}

// 8-bit low part of the pair form: x = hi + (b - 128) * ulp(hi) / 256 with the byte b = round((x - hi) * 256 / ulp(hi)) + 128
// in [0, 255] (|x - hi| <= ulp / 2), ulp(hi) = 2^(E - 134) for hi's biased exponent E: x is held to ulp / 512 = 2^-17 relative,
// the bf16 low part's accuracy, in one byte.  Written for the VALU budget -- at 8 bytes per element the kernel would otherwise be
// instruction-bound (0.5 T elements/s x ~40 operations against 33 T lane-operations/s): the scale factors are exponent
// arithmetic (v_bfe_u32 + v_ldexp_f32, no table, no division), the byte leaves through v_cvt_pk_u8_f32 (saturating convert
// AND insert) and arrives through v_cvt_f32_ubyteN.  |hi| < 2^-111 (E < 16): the low part under- / overflows harmlessly.
__device__ __forceinline__ uint32_t lo8_insert(float x, float h, uint32_t E, uint32_t byte_idx, uint32_t acc) {
    return __builtin_amdgcn_cvt_pk_u8_f32(ldexpf(x - h, 142 - (int)E) + 128.0f, byte_idx, acc);  // (the convert rounds to nearest)
}
__device__ __forceinline__ float lo8_value(float byte_as_float, uint32_t E) { return ldexpf(byte_as_float - 128.0f, (int)E - 142); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// attention_pipe.hip: persistent, LDS-DMA-pipelined window attention on pre-normalised bf16 q/k (head_dim 88)
struct AttnPipeArgs {
    const void* qkv;
    void* out;
    int64_t ldq, ldo;
    int B, gh, gw, heads, sh, sw, dbg;
    const float* scale;  // per-head logit scale parameter (bounds |logit|); null = unknown
    int tiled;           // qkv is window-tiled: [sample][window][head][q|k|v][256][hd] (SWIFTK_ATTN_TILED)
    int hd;              // head_dim: 80, 88 or 96
};
int swiftk_launch_attn_pipe(const AttnPipeArgs& a, hipStream_t st);

// Zero n floats with an ORDINARY kernel on the caller's stream.  Not hipMemsetAsync: captured into a HIP graph and replayed on the
// null stream, a memset writes a STALE fill pattern under the HIP 7.0.x runtime PyTorch bundles (the node does not own its pattern:
// zeros turn into later launches' kernel arguments) -- what overflowed swiftk_modnorm_bwd's atomically accumulated column sums in
// round 5 (standalone: tools/memset_graph_repro.hip).  A kernel node carries its arguments by value.
__global__ __launch_bounds__(256) static void swiftk_zero_f32_kernel(float* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}
static inline int swiftk_zero_f32_launch(float* p, int64_t n, hipStream_t st) {
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(swiftk_zero_f32_kernel, dim3((unsigned)(blocks < 1024 ? (blocks < 1 ? 1 : blocks) : 1024)), dim3(256), 0, st, p, n);
    return (int)hipGetLastError();
}

int swiftk_zero_f32_impl(float* p, int64_t n, void* stream, int who);  // (elementwise.hip; who = 1: a clear inside the library, tuning key 25)
extern int g_zero_memset;  // tuning key 25
int swiftk_zero_check_enable();
int swiftk_zero_check_launch(const float* p, int64_t n, void* stream);  // (no-op unless key 25 bit 4 was set)

// live per-kernel timing (bench.py's roofline legs; state lives in gemm.hip): a launch of kind `kind` (a GEMM
// epilogue code, or SWIFTK_PROF_ATTENTION) with matching n is bracketed by HIP events on its own stream
bool swiftk_prof_begin(int kind, int n, hipStream_t st);
void swiftk_prof_end(hipStream_t st);
extern int g_fwd_tiled;  // tuning key 5
extern int g_fwd_fused;  // tuning key 8
extern int g_x3_exact;  // tuning key 11
extern int g_fwd_pair;  // tuning key 12
extern int g_f32_chunk_k;  // tuning key 13
extern int g_fwd_splitk;  // tuning key 14
extern int g_attn_bwd_pipe;  // tuning key 9
extern int g_attn_bwd_fuse;  // tuning key 15
extern int g_modnorm_bwd_fused;  // tuning key 16
extern int g_modnorm_jvp_rows;  // tuning key 17
extern int g_x3_ffsplit;  // tuning key 18
extern int g_x3_normsplit;  // tuning key 26
extern int g_x3_qkonly;  // tuning key 27
extern int g_x3_attnpv;  // tuning key 28
extern int g_fwd_tail;   // tuning key 29 (forward.hip)
extern int g_fwd_pepair;  // tuning key 19
extern int g_modnorm_nt;  // tuning key 6
extern int g_persist_wgs;  // tuning key 2: workgroups of the persistent matrix kernels (GEMM, fused to_qkv + attention)
extern int g_fwd_rownorm;  // tuning key 23: wo / w2 + norm as one complete-row kernel up to this many units per step
extern int g_rownorm_dbg;
extern int g_tn_pp;  // tuning key 22: ping-pong k-loop of gemm_tn_kernel
extern int g_attn_pp;  // tuning key 21: ping-pong k-loop of the fused to_qkv + attention kernel
extern int g_attn_dbg;  // tuning key 4: attention ablation bits (timing experiments only)
