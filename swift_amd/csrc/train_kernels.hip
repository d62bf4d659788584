// Backward-pass and loss kernels of the training step (reference src/swift/training/{trainer,loss}.py) for gfx950.
// All HBM-bound; the MFMA work of the backward pass reuses gemm.hip (dgrad: A = dY, W = W^T copy; wgrad: A = dY^T,
// W = X^T with split-K fp32 slabs) and attention_bwd.hip.
#include "common.h"

namespace {

inline int grid_for(int64_t work_items, int per_block = 256, int cap = 256 * 16) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename T>
__device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void st4(T* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void st4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(a, b), pack_bf16(c, d));
}

// ------------------------------------------------------------------------------------------ transpose
// dst[c][r] = src[r][c], 64x64 tiles through LDS (+1 padding), columns [rows, ldd) of dst zero-filled.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, int64_t lds, T* __restrict__ dst,
                                                        int64_t ldd, int64_t rows, int64_t cols) {
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < rows && c < cols) ? src[r * lds + c] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < ldd) dst[c * ldd + r] = tile[tx][ty + 4 * i];
    }
}

// bf16 fast path: 64x64 tiles, 16-B global loads and stores on both sides (every tile row is one 128-B line), the
// transposition itself as 2-B LDS writes into a [64][72] image (144-B rows keep the 16-B read-back aligned).
// Needs 16-B aligned bases and lds, ldd multiples of 8.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, int64_t lds, bf16_t* __restrict__ dst,
                                                             int64_t ldd, int64_t rows, int64_t cols) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 72];
    const int t = threadIdx.x, ch = t & 7, rr = t >> 3;  // 8 chunks x 32 rows per pass
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = rr + 32 * i;
        const int64_t r = r0 + row, c = c0 + 8 * ch;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < rows && c + 8 <= cols) {
            v = *reinterpret_cast<const uint4*>(src + r * lds + c);
        } else if (r < rows) {
            bf16_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = c + k < cols ? src[r * lds + c + k] : bf16_t(0);
            v = *reinterpret_cast<uint4*>(e);
        }
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[(8 * ch + k) * 72 + row] = e[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = rr + 32 * i;  // row of the transposed tile
        const int64_t c = c0 + col, r = r0 + 8 * ch;
        if (c < cols && r < ldd) {
            const uint4 v = *reinterpret_cast<const uint4*>(tile + col * 72 + 8 * ch);  // rows >= `rows` were zero-filled above
            *reinterpret_cast<uint4*>(dst + c * ldd + r) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ slab reduction
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int64_t ld_slab,
                                                           int64_t slab_stride, int nslabs, float* __restrict__ out,
                                                           int64_t ld_out, int64_t rows, int64_t cols, int accumulate) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols, c = i - r * cols;
        float s = accumulate ? out[r * ld_out + c] : 0.f;
        for (int k = 0; k < nslabs; ++k) s += slabs[k * slab_stride + r * ld_slab + c];
        out[r * ld_out + c] = s;
    }
}

// ------------------------------------------------------------------------------------------ SwiGLU (unfused form)
// h [M, 2*mlp] with columns interleaved (gate_j, up_j) -> hmid[m][j] = silu(gate) * up   (swinv2.py:99-100)
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* __restrict__ h, int64_t ldh, T* __restrict__ o, int64_t ldo,
                                                         int64_t M, int mlp) {
    const int64_t total = M * mlp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / mlp;
        const int j = (int)(i - m * mlp);
        const float g = ldf(h + m * ldh + 2 * j), u = ldf(h + m * ldh + 2 * j + 1);
        o[m * ldo + j] = elem<T>::from_f(g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)) * u);
    }
}
// dh[m][2j] = dho * up * (s + g s (1-s)),  dh[m][2j+1] = dho * g s,   s = sigmoid(gate)
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ h, int64_t ldh, const T* __restrict__ dho,
                                                         int64_t ldo, T* __restrict__ dh, int64_t lddh, int64_t M, int mlp) {
    const int64_t total = M * mlp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / mlp;
        const int j = (int)(i - m * mlp);
        const float g = ldf(h + m * ldh + 2 * j), u = ldf(h + m * ldh + 2 * j + 1), d = ldf(dho + m * ldo + j);
        const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
        dh[m * lddh + 2 * j] = elem<T>::from_f(d * u * (s + g * s * (1.0f - s)));
        dh[m * lddh + 2 * j + 1] = elem<T>::from_f(d * g * s);
    }
}

// ------------------------------------------------------------------------------------------ modnorm backward
// out = LN(y; gamma, beta) * (1 + sc_b) + sh_b,  upstream g = dL/dout (fp32 [M, d]).
//   dy = rstd * (dn - mean(dn) - n * mean(dn * n)),  dn = g (1+sc) gamma,  n = (y - mu) rstd
//   dgamma += sum_rows g (1+sc) n,  dbeta += sum_rows g (1+sc),  dsc_b += sum_rows g ln,  dsh_b += sum_rows g
// Two streaming passes instead of one serial one: (A) one wave per row -- statistics, dy, and (mu, rstd) saved per row;
// (B) column sums with a lane per four columns walking rows (no cross-lane reduction: the statistics are known), four
// waves of a block on four row sub-ranges, combined in LDS, one atomic per column and output per block.
template <typename T>
__global__ __launch_bounds__(256) void modnorm_bwd_rows_kernel(const T* __restrict__ y, int64_t ldy, const float* __restrict__ g,
                                                               T* __restrict__ dy, int64_t lddy, const float* __restrict__ gamma,
                                                               const float* __restrict__ mod, int64_t ldmod,
                                                               float* __restrict__ stats, int64_t M, int d, int64_t rps,
                                                               float eps) {
    constexpr int SLOTS = 6;  // d <= 1536
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nc = d >> 2;
    const float* mrow = mod + (row / rps) * ldmod;
    float v[SLOTS][4], dn[SLOTS][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            float gg[4], ga[4], sc[4];
            ld4<T>(y + row * ldy + 4 * c, v[i]);
            ld4<float>(g + row * d + 4 * c, gg);
            ld4<float>(gamma + 4 * c, ga);
            ld4<float>(mrow + 4 * c, sc);
#pragma unroll
            for (int e = 0; e < 4; ++e) dn[i][e] = gg[e] * (1.0f + sc[e]) * ga[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = dn[i][e] = 0.f;
        }
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float inv_d = 1.0f / (float)d;
    const float mean = wave_sum(sum) * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = (lane + 64 * i < nc) ? v[i][e] - mean : 0.f;
            v[i][e] = t;
            sq += t * t;
        }
    const float rstd = rsqrtf(wave_sum(sq) * inv_d + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[i][e] *= rstd;
            s1 += dn[i][e];
            s2 += dn[i][e] * v[i][e];
        }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = lane + 64 * i;
        if (c < nc)
            st4<T>(dy + row * lddy + 4 * c, rstd * (dn[i][0] - s1 - v[i][0] * s2), rstd * (dn[i][1] - s1 - v[i][1] * s2),
                   rstd * (dn[i][2] - s1 - v[i][2] * s2), rstd * (dn[i][3] - s1 - v[i][3] * s2));
    }
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// Both passes in one (round 4): a block owns `rows_per_block` consecutive rows of ONE sample, its four waves take every fourth row.
// The column sums need only two running sums per column,
//     P1 = sum_rows g n,   P2 = sum_rows g:    dgamma += (1+sc) P1,  dbeta += (1+sc) P2,  dsc_b += gamma P1 + beta P2,  dsh_b += P2,
// held per lane for its columns (2 x 4 x SLOTS registers) next to the row constant w = (1+sc) gamma the row pass multiplies by;
// y and g are read once (8 instead of 14 bytes per element over the two kernels above), the four waves' sums meet in LDS and leave
// as one atomic per column and output.
template <typename T>
struct Raw4;  // four consecutive elements as loaded
template <>
struct Raw4<float> {
    typedef float4 type;
    __device__ static void cvt(const float4& r, float (&v)[4]) { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
};
template <>
struct Raw4<bf16_t> {
    typedef uint2 type;
    __device__ static void cvt(const uint2& r, float (&v)[4]) {
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
};

template <typename T, int SLOTS>
__global__ __launch_bounds__(256) void modnorm_bwd_fused_kernel(const T* __restrict__ y, int64_t ldy, const float* __restrict__ g,
                                                                T* __restrict__ dy, int64_t lddy, const float* __restrict__ gamma,
                                                                const float* __restrict__ mod, int64_t ldmod, float* __restrict__ psum,
                                                                int64_t M, int d, int64_t rps, float eps, int rows_per_block) {
    typedef typename Raw4<T>::type raw_t;
    constexpr int dpad = SLOTS * 256;
    __shared__ __attribute__((aligned(16))) float red[8 * dpad];  // [wave][P1 | P2][column]
    __shared__ __attribute__((aligned(16))) float wl[dpad];       // the row constant w = (1 + sc) gamma
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, M);
    const int64_t b = r0 / rps;
    const int nc = d >> 2;
    const float* mrow = mod + b * ldmod;
    for (int col = threadIdx.x; col < dpad; col += 256) wl[col] = col < d ? (1.0f + mrow[col]) * gamma[col] : 0.f;
    float p1[SLOTS][4], p2[SLOTS][4];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) p1[i][e] = p2[i][e] = 0.f;
    __syncthreads();
    const float inv_d = 1.0f / (float)d;
    // the next row's loads are in flight while this row is reduced and stored (a lone wave took 3.6 us per row without)
    raw_t ry[SLOTS];
    float4 rg[SLOTS];
    auto fetch = [&](int64_t row) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                ry[i] = *reinterpret_cast<const raw_t*>(y + row * ldy + 4 * c);
                rg[i] = *reinterpret_cast<const float4*>(g + row * d + 4 * c);
            }
        }
    };
    if (r0 + wv < r1) fetch(r0 + wv);
    for (int64_t row = r0 + wv; row < r1; row += 4) {
        float v[SLOTS][4], gg[SLOTS][4];
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            if (lane + 64 * i < nc) {
                Raw4<T>::cvt(ry[i], v[i]);
                gg[i][0] = rg[i].x; gg[i][1] = rg[i].y; gg[i][2] = rg[i].z; gg[i][3] = rg[i].w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = gg[i][e] = 0.f;
            }
        }
        if (row + 4 < r1) fetch(row + 4);
        // ONE round of wave reductions per row: with t = y - y0 (y0 = the row's first element: a shift that keeps the one-pass
        // variance well conditioned) the four sums  sum t, sum t^2, sum dn, sum dn t  give mean, rstd and both means the row's dy
        // needs (mean(dn), mean(dn n) = rstd (sum dn t - mean_t sum dn) / d) -- not four dependent reductions
        const float y0 = __shfl(v[0][0], 0, 64);
        float q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const float4 w4 = *reinterpret_cast<const float4*>(&wl[4 * (lane + 64 * i)]);
            const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (lane + 64 * i < nc) ? v[i][e] - y0 : 0.f;
                const float dn = gg[i][e] * ww[e];
                v[i][e] = t;
                q1 += t;
                q2 = fmaf(t, t, q2);
                q3 += dn;
                q4 = fmaf(dn, t, q4);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            q1 += __shfl_xor(q1, o, 64);
            q2 += __shfl_xor(q2, o, 64);
            q3 += __shfl_xor(q3, o, 64);
            q4 += __shfl_xor(q4, o, 64);
        }
        const float mt = q1 * inv_d;
        const float rstd = rsqrtf(fmaxf(q2 * inv_d - mt * mt, 0.f) + eps);
        const float s1 = q3 * inv_d, s2 = rstd * (q4 - mt * q3) * inv_d;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const float4 w4 = *reinterpret_cast<const float4*>(&wl[4 * (lane + 64 * i)]);
            const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
            float o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float n = (v[i][e] - mt) * rstd;  // (pad lanes: g = 0, nothing is added or stored)
                p1[i][e] = fmaf(gg[i][e], n, p1[i][e]);
                p2[i][e] += gg[i][e];
                o4[e] = rstd * (gg[i][e] * ww[e] - s1 - n * s2);
            }
            const int c = lane + 64 * i;
            if (c < nc) st4<T>(dy + row * lddy + 4 * c, o4[0], o4[1], o4[2], o4[3]);
        }
    }
    // the four waves' sums meet in LDS indexed by COLUMN, so that the atomics below run over consecutive addresses (whole lines
    // per instruction; issued in the accumulators' own layout -- a lane's four columns 16 B apart -- they were a quarter-line each
    // and cost more than the pass they replaced)
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            *reinterpret_cast<float4*>(&red[(wv * 2 + 0) * dpad + 4 * c]) = make_float4(p1[i][0], p1[i][1], p1[i][2], p1[i][3]);
            *reinterpret_cast<float4*>(&red[(wv * 2 + 1) * dpad + 4 * c]) = make_float4(p2[i][0], p2[i][1], p2[i][2], p2[i][3]);
        }
    }
    __syncthreads();
    // per SAMPLE sums first (psum [2][samples][d], zeroed by the launcher): every block of the launch adding into dgamma / dbeta
    // directly put 1,024 atomics on each of their 2 x d addresses, which cost more than the pass this kernel removes
    const int64_t nb = M / rps;
    for (int col = threadIdx.x; col < d; col += 256) {
        const float a1 = (red[0 * dpad + col] + red[2 * dpad + col]) + (red[4 * dpad + col] + red[6 * dpad + col]);
        const float a2 = (red[1 * dpad + col] + red[3 * dpad + col]) + (red[5 * dpad + col] + red[7 * dpad + col]);
        atomicAdd(psum + b * d + col, a1);
        atomicAdd(psum + (nb + b) * d + col, a2);
    }
}

// ... and the outputs from the per-sample sums: one thread per column
__global__ __launch_bounds__(256) void modnorm_bwd_finish_kernel(float* __restrict__ psum, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ mod,
                                                                 int64_t ldmod, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ dmod, int64_t lddmod, int nb, int d) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= d) return;
    const float ga = gamma[col], be = beta[col];
    float dg = 0.f, db = 0.f;
    for (int b = 0; b < nb; ++b) {
        const float a1 = psum[(int64_t)b * d + col], a2 = psum[(int64_t)(nb + b) * d + col];
        // the sums are consumed: leave the workspace zero for the next call (the launcher clears it itself only when it cannot
        // know that -- first use, or after the two-kernel form kept row statistics there)
        psum[(int64_t)b * d + col] = 0.f;
        psum[(int64_t)(nb + b) * d + col] = 0.f;
        const float sc1 = 1.0f + mod[b * ldmod + col];
        dg = fmaf(sc1, a1, dg);
        db = fmaf(sc1, a2, db);
        dmod[b * lddmod + col] += ga * a1 + be * a2;
        dmod[b * lddmod + d + col] += a2;
    }
    dgamma[col] += dg;
    dbeta[col] += db;
}

template <typename T>
__global__ __launch_bounds__(256) void modnorm_bwd_cols_kernel(const T* __restrict__ y, int64_t ldy, const float* __restrict__ g,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ mod, int64_t ldmod,
                                                               const float* __restrict__ stats, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dmod,
                                                               int64_t lddmod, int d, int64_t rps, int rows_per_block) {
    __shared__ float red[3][64][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;  // float4 column slot
    const bool ok = c < (d >> 2);
    const int64_t blocks_per_sample = (rps + rows_per_block - 1) / rows_per_block;
    const int64_t b = blockIdx.y / blocks_per_sample;
    const int64_t r0 = b * rps + (blockIdx.y - b * blocks_per_sample) * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, (b + 1) * rps);
    float ga[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        ld4<float>(gamma + 4 * c, ga);
        ld4<float>(beta + 4 * c, be);
        ld4<float>(mod + b * ldmod + 4 * c, sc);
    }
    float acc[16];  // dg[4], db[4], ds[4], dh[4]
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    if (ok) {
        for (int64_t row = r0 + wv; row < r1; row += 4) {
            float v[4], gg[4];
            ld4<T>(y + row * ldy + 4 * c, v);
            ld4<float>(g + row * d + 4 * c, gg);
            const float mean = stats[2 * row], rstd = stats[2 * row + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float n = (v[e] - mean) * rstd;
                const float dl = gg[e] * (1.0f + sc[e]);
                acc[e] += dl * n;
                acc[4 + e] += dl;
                acc[8 + e] += gg[e] * (n * ga[e] + be[e]);
                acc[12 + e] += gg[e];
            }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) red[wv - 1][lane][k] = acc[k];
    }
    __syncthreads();
    if (wv == 0 && ok) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] += red[0][lane][k] + red[1][lane][k] + red[2][lane][k];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(dgamma + 4 * c + e, acc[e]);
            atomicAdd(dbeta + 4 * c + e, acc[4 + e]);
            atomicAdd(dmod + b * lddmod + 4 * c + e, acc[8 + e]);
            atomicAdd(dmod + b * lddmod + d + 4 * c + e, acc[12 + e]);
        }
    }
}

// ------------------------------------------------------------------------------------------ QK-norm backward
// Forward (SWIFTK_EPI_QKNORM): qh = tau_h * q/|q|, kh = k/|k|, v unchanged; rn = 1/max(|.|, 1e-12) saved per vector.
//   dq = tau rn (dqh - u (u . dqh)), u = qh/tau;   dk = rn (dkh - kh (kh . dkh));   dv = dvh
//   d scale_h += tau * sum_tokens (u . dqh)   (zero when the clamp at ln 100 is active)
// Sixteen lanes per (token, head vector), one 16-B chunk per lane (10 / 11 / 12 of them live for head_dim 80 / 88 / 96), so a
// wave's four vectors are one contiguous 4 x 2*head_dim-byte run of the row: whole-line requests (a lane per vector, each
// walking its own 176 B, made 11 requests of 16 scattered bytes per lane and ran at 60 % of this form's rate).  The dot
// product is a 4-step butterfly inside the 16-lane group; UNR vectors per group are in flight at once.
template <typename T>
__global__ __launch_bounds__(256) void qknorm_bwd_kernel(const T* __restrict__ qkvh, const T* dqkvh,
                                                         int64_t ld, const float* __restrict__ rn, T* dqkv,
                                                         int64_t ldo, const float* __restrict__ scale,
                                                         float* __restrict__ dscale, int64_t M, int heads, int hd, int inplace) {
    constexpr int PER = 16 / (int)sizeof(T);  // elements per 16-B chunk
    constexpr int UNR = 4;
    __shared__ float sacc[64];
    if (threadIdx.x < 64) sacc[threadIdx.x] = 0.f;
    __syncthreads();
    // in place (dqkv == dqkvh, ldo == ld: the attention backward wrote straight into the GEMM operand buffer): v's gradient is
    // already where it belongs, only the q-hat / k-hat vectors are walked
    const int nvec = (inplace ? 2 : 3) * heads, nch = hd / PER;
    const int64_t total = M * nvec;
    const int sub = threadIdx.x & 15;                                   // chunk of the vector
    const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;  // 16-lane group
    const int64_t ngrp = ((int64_t)gridDim.x * 256) >> 4;
    const bool live = sub < nch;
    for (int64_t i0 = grp; i0 < total; i0 += ngrp * UNR) {
        uint4 ra[UNR], rd[UNR];
        float r[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + u * ngrp;
            ra[u] = rd[u] = make_uint4(0u, 0u, 0u, 0u);
            r[u] = 0.f;
            if (i < total && live) {
                const int64_t m = i / nvec;
                const int vv = (int)(i - m * nvec);
                const int v = inplace ? (vv >> 1) * 3 + (vv & 1) : vv;
                const int64_t col = (int64_t)v * hd + sub * PER;
                rd[u] = *reinterpret_cast<const uint4*>(dqkvh + m * (inplace ? ldo : ld) + col);
                if (v % 3 != 2) {
                    ra[u] = *reinterpret_cast<const uint4*>(qkvh + m * ld + col);
                    r[u] = rn[m * (3 * heads) + v];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + u * ngrp;
            if (i >= total) break;  // (group-uniform)
            const int64_t m = i / nvec;
            const int vv = (int)(i - m * nvec);
            const int v = inplace ? (vv >> 1) * 3 + (vv & 1) : vv, kind = v % 3, h = v / 3;
            T* o = dqkv + m * ldo + (int64_t)v * hd + sub * PER;
            if (kind == 2) {
                if (live) *reinterpret_cast<uint4*>(o) = rd[u];
                continue;
            }
            const T* pa = reinterpret_cast<const T*>(&ra[u]);
            const T* pd = reinterpret_cast<const T*>(&rd[u]);
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < PER; ++e) dot += elem<T>::to_f(pa[e]) * elem<T>::to_f(pd[e]);
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) dot += __shfl_xor(dot, sft, 16);
            const float s = scale[h];
            const float tau = kind == 0 ? expf(fminf(s, 4.605170185988092f)) : 1.0f;
            const float itau = 1.0f / tau;
            dot *= itau;  // u . d(qh)   (a = tau u)
            const float f = tau * r[u];
            uint4 ov;
            T* po = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int e = 0; e < PER; ++e) po[e] = elem<T>::from_f(f * (elem<T>::to_f(pd[e]) - elem<T>::to_f(pa[e]) * itau * dot));
            if (live) *reinterpret_cast<uint4*>(o) = ov;
            if (kind == 0 && s < 4.605170185988092f && sub == 0) atomicAdd(&sacc[h], tau * dot);  // LDS first: 12 hot addresses
        }
    }
    __syncthreads();
    if (threadIdx.x < heads && sacc[threadIdx.x] != 0.f) atomicAdd(dscale + threadIdx.x, sacc[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------ column sums
// out[c] (+)= sum_r src[r][c]   (bias / pos_embed gradients); pos: out[(r % period)][c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ out,
                                                     int64_t rows, int cols, int64_t period, int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    for (int c = threadIdx.x; c < cols; c += 256) {
        if (period > 0) {
            for (int64_t r = r0; r < min(r0 + rows_per_block, rows); ++r) atomicAdd(out + (r % period) * cols + c, src[r * lds + c]);
        } else {
            float s = 0.f;
            for (int64_t r = r0; r < min(r0 + rows_per_block, rows); ++r) s += src[r * lds + c];
            atomicAdd(out + c, s);
        }
    }
}

// The patch embedding's three consumers of d(x0) [samples x period rows, cols] in ONE pass: bias[c] += sum over all rows,
// pos[t][c] += sum over the samples of row (b period + t) -- a thread owns (t, c), so no atomics and a fixed summation order
// -- and the bf16 copy (row padding zeroed) the weight-gradient GEMM reads.  A block takes 8 consecutive tokens.
__global__ __launch_bounds__(256) void embed_bwd_sums_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ bias,
                                                             float* __restrict__ pos, bf16_t* __restrict__ dst, int64_t ldd,
                                                             int nsamp, int64_t period, int cols) {
    constexpr int TB = 8;  // tokens per block (must match the launcher)
    const int64_t t0 = (int64_t)blockIdx.x * TB;
    for (int c = threadIdx.x; c < (dst ? (int)ldd : cols); c += 256) {
        float sb = 0.f;
        for (int64_t t = t0; t < min(t0 + TB, period); t += 2) {
            float s[2] = {0.f, 0.f};
            const bool two = t + 1 < period;
            for (int b0 = 0; b0 < nsamp; b0 += 8) {  // two tokens x eight samples: sixteen loads in flight before anything is stored
                float v[2][8];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        v[u][j] = (c < cols && b0 + j < nsamp && (u == 0 || two)) ? src[((int64_t)(b0 + j) * period + t + u) * lds + c] : 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        s[u] += v[u][j];
                        if (dst && b0 + j < nsamp && (u == 0 || two)) dst[((int64_t)(b0 + j) * period + t + u) * ldd + c] = f2bf(v[u][j]);
                    }
            }
            if (c < cols) {
                pos[t * cols + c] += s[0];
                if (two) pos[(t + 1) * cols + c] += s[1];
            }
            sb += s[0] + s[1];
        }
        if (c < cols) atomicAdd(bias + c, sb);
    }
}

// ------------------------------------------------------------------------------------------ small linear backward
// y[b][n] = act(x[b][:] . W[n][:] + bias[n]) with B <= 64 (time-embedding MLP, modulation, logvar)
//   dz = dy * act'(z) (act = SiLU needs y's pre-activation z; we recompute from y via saved z),
//   dx[b][k] = sum_n dz[b][n] W[n][k],  dW[n][k] += sum_b dz[b][n] x[b][k],  dbias[n] += sum_b dz[b][n]
__global__ __launch_bounds__(256) void small_dgrad_kernel(const float* __restrict__ dz, int64_t lddz,
                                                          const float* __restrict__ W, int64_t ldw, float* __restrict__ dx,
                                                          int64_t lddx, int B, int N, int K, int n_chunk) {
    // block (k-slab of 256, n-chunk): partial sums over its n range, atomically added.  W (N x K fp32: 214 MB for the
    // modulation Linears of Swift-B) is streamed ONCE per group of eight samples -- eight accumulators per thread, the dz
    // values are wave-uniform loads -- so the kernel runs at the HBM rate of one pass over W.
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n0 = blockIdx.y * n_chunk, n1 = min(n0 + n_chunk, N);
    if (k >= K) return;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int nb = min(8, B - b0);
        // (sixteen weight rows' loads in flight per thread: with four the kernel ran at a third of the HBM rate)
        for (int nn = n0; nn < n1; nn += 16) {
            float w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = nn + i < n1 ? W[(int64_t)(nn + i) * ldw + k] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (nn + i < n1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j < nb) s[j] += dz[(int64_t)(b0 + j) * lddz + nn + i] * w[i];
                }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nb) atomicAdd(dx + (int64_t)(b0 + j) * lddx + k, s[j]);
    }
}
template <int WG_ROWS>
__global__ __launch_bounds__(256) void small_wgrad_kernel(const float* __restrict__ dz, int64_t lddz,
                                                          const float* __restrict__ x, int64_t ldx, float* __restrict__ dW,
                                                          int64_t lddw, float* __restrict__ dbias, int B, int N, int K) {
    // a block owns WG_ROWS consecutive rows n of dW (the modulation Linears: 50,688 rows of 4 KB -- one row per block left the
    // kernel bound by block turnover): a thread keeps the samples' x[b][k] in registers for eight samples at a time and
    // updates its k of every row, so eight independent read-modify-writes are in flight per thread
    const int n0 = blockIdx.x * WG_ROWS, nr = min(WG_ROWS, N - n0);
    // the block's dz values (eight samples x eight rows per round) go through LDS once: read per use as wave-uniform global
    // loads they made every FMA wait for a scalar load
    __shared__ float sdz[8][WG_ROWS];
    float s[5][WG_ROWS];  // this thread's k = threadIdx.x + 256 i, i < 5 (K <= 1280; wider K: the outer loop)
    for (int kb = 0; kb < K; kb += 5 * 256) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int r = 0; r < WG_ROWS; ++r) s[i][r] = 0.f;
        for (int b0 = 0; b0 < B; b0 += 8) {
            __syncthreads();
            if (threadIdx.x < 8 * WG_ROWS) {
                const int j = threadIdx.x / WG_ROWS, r = threadIdx.x % WG_ROWS;
                sdz[j][r] = (b0 + j < B && r < nr) ? dz[(int64_t)(b0 + j) * lddz + n0 + r] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int k = kb + threadIdx.x + 256 * i;
                if (k < K) {
                    float xv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = b0 + j < B ? x[(int64_t)(b0 + j) * ldx + k] : 0.f;
#pragma unroll
                    for (int r = 0; r < WG_ROWS; ++r)
#pragma unroll
                        for (int j = 0; j < 8; ++j) s[i][r] += sdz[j][r] * xv[j];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = kb + threadIdx.x + 256 * i;
            if (k < K) {
#pragma unroll
                for (int r = 0; r < WG_ROWS; ++r)
                    if (r < nr) dW[(int64_t)(n0 + r) * lddw + k] += s[i][r];
            }
        }
    }
    if (dbias && threadIdx.x < nr) {
        float db = 0.f;
        for (int b = 0; b < B; ++b) db += dz[(int64_t)b * lddz + n0 + threadIdx.x];
        dbias[n0 + threadIdx.x] += db;
    }
}
// dz = dy * silu'(z): silu'(z) = s (1 + z (1 - s))
__global__ void silu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy, float* __restrict__ dz, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = 1.0f / (1.0f + expf(-z[i]));
    dz[i] = dy[i] * s * (1.0f + z[i] * (1.0f - s));
}

// ------------------------------------------------------------------------------------------ losses
// almost-fair CRPS over an ensemble of m members (training/loss.py:343-371), weighted and reduced:
//   loss = 1/(B H W) sum_{b,c,h,w} w_var[c] w_lat[h] ( mean_i |x_i - y| - (1-eps)/(2 m (m-1)) sum_{i != j} |x_i - x_j| )
// preds [m][B,C,H,W]; dpreds (same layout, optional) receives dloss/dpreds * gscale.
__global__ __launch_bounds__(256) void crps_kernel(const float* __restrict__ preds, const float* __restrict__ target,
                                                   const float* __restrict__ w_var, const float* __restrict__ w_lat,
                                                   float* __restrict__ loss, float* __restrict__ dpreds, int m, int64_t n,
                                                   int C, int H, int W, float alpha, float gscale, float inv_bhw) {
    const float eps = (1.0f - alpha) / (float)m;
    const float cs = (1.0f - eps) / (2.0f * m * (m - 1));
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int h = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
        const float w = w_var[c] * w_lat[h];
        const float y = target[i];
        float skill = 0.f, spread = 0.f;
        for (int a = 0; a < m; ++a) {
            const float xa = preds[(int64_t)a * n + i];
            skill += fabsf(xa - y);
            float ga = 0.f;
            for (int b2 = 0; b2 < m; ++b2) {
                const float xb = preds[(int64_t)b2 * n + i];
                spread += fabsf(xa - xb);
                ga += (xa > xb) - (xa < xb);
            }
            if (dpreds) {
                const float sg = (float)((xa > y) - (xa < y));
                dpreds[(int64_t)a * n + i] = gscale * inv_bhw * w * (sg / m - cs * 2.0f * ga);
            }
        }
        acc += w * (skill / m - cs * spread);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv_bhw);
}

// TrigFlow (training/loss.py:132-160): prep  x_t/sd = (cos t x + sin t sd z)/sd,  v_t = cos t sd z - sin t x
__global__ __launch_bounds__(256) void trigflow_prep_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                            const float* __restrict__ t, float* __restrict__ xt_over_sd,
                                                            float* __restrict__ vt, float sd, int64_t per_sample, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float tt = t[i / per_sample], c = cosf(tt), s = sinf(tt);
        const float zz = z[i] * sd;
        xt_over_sd[i] = (c * x[i] + s * zz) / sd;
        vt[i] = c * zz - s * x[i];
    }
}
// loss = 1/(B H W) sum [ exp(-lv_b) w (sd F - v)^2 + lv_b ];  dF, dlv optional
__global__ __launch_bounds__(256) void trigflow_loss_kernel(const float* __restrict__ F, const float* __restrict__ vt,
                                                            const float* __restrict__ logvar, const float* __restrict__ w_var,
                                                            const float* __restrict__ w_lat, float* __restrict__ loss,
                                                            float* __restrict__ dF, float* __restrict__ dlogvar, float sd,
                                                            int64_t n, int C, int H, int W, float gscale, float inv_bhw) {
    float acc = 0.f;
    const int64_t per_sample = (int64_t)C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int h = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t b = i / per_sample;
        const float lv = logvar ? logvar[b] : 0.f, iv = expf(-lv);
        const float w = w_var[c] * w_lat[h];
        const float r = sd * F[i] - vt[i];
        acc += iv * w * r * r + lv;
        if (dF) dF[i] = gscale * inv_bhw * 2.0f * sd * iv * w * r;
        if (dlogvar) atomicAdd(dlogvar + b, gscale * inv_bhw * (1.0f - iv * w * r * r));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv_bhw);
}

// out = a[b] * x + c[b] * y  (per-sample coefficients, fp32)
__global__ __launch_bounds__(256) void axpby_ps_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                       const float* __restrict__ x, const float* __restrict__ c,
                                                       const float* __restrict__ y, int64_t per_sample, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_sample;
        out[i] = a[b] * x[i] + (y ? c[b] * y[i] : 0.f);
    }
}

// cond update of the multistep loss / rollout with gradient: x_next_std = x_std + y * (st/sx)   and its transpose
__global__ __launch_bounds__(256) void chan_axpy_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                        const float* __restrict__ y, const float* __restrict__ coef, int C,
                                                        int64_t hw, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i / hw) % C);
        out[i] = (x ? x[i] : 0.f) + coef[c] * y[i];
    }
}

// ------------------------------------------------------------------------------------------ validation RMSE sums
// sq[0] += sum (y - t)^2 ;  sq[1 + c] += sum_{b,h,w} w_lat[h] (y - t)^2   (training/validate.py:96-107; the caller divides
// and takes the roots).  One block per (channel, chunk): block-reduced, two atomics per block.
__global__ __launch_bounds__(256) void rmse_sums_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                        int64_t t_batch_stride, const float* __restrict__ w_lat,
                                                        float* __restrict__ sq, int B, int C, int H, int W) {
    __shared__ float red[2][4];
    const int c = blockIdx.y;
    const int64_t hw = (int64_t)H * W, per = (int64_t)B * hw;
    float a0 = 0.f, a1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hw, r = i - b * hw;
        const int h = (int)(r / W);
        const float d = y[(b * C + c) * hw + r] - t[b * t_batch_stride + c * hw + r];
        a0 += d * d;
        a1 += w_lat[h] * d * d;
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a0;
        red[1][threadIdx.x >> 6] = a1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sq, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(sq + 1 + c, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// ------------------------------------------------------------------------------------------ ensemble metric sums
// Per (sample b, variable v), over the grid with latitude weights (eval/metrics.py:39-134):
//   out[b][v][0] = sum w (mean_n x - y)^2          (ensemble-mean RMSE)
//   out[b][v][1] = sum_n sum w |x_n - y|           (CRPS skill term)
//   out[b][v][2] = sum_{n,n'} sum w |x_n - x_n'|   (CRPS spread term)
//   out[b][v][3] = sum w var_n(x), unbiased        (spread of the spread/skill ratio)
// One block per (b, v, chunk of the grid); the N member values of a grid point live in registers (N <= 64).
template <int NMAX>
__global__ __launch_bounds__(256) void ensemble_sums_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                            const float* __restrict__ w_lat, float* __restrict__ out, int N,
                                                            int V, int H, int W) {
    __shared__ float red[4][4];
    const int bv = blockIdx.y, b = bv / V, v = bv - b * V;
    const int64_t hw = (int64_t)H * W;
    const float* p0 = pred + ((int64_t)b * N * V + v) * hw;  // member stride V*hw
    const float* yy = y + (int64_t)bv * hw;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
        const float w = w_lat[i / W], t = yy[i];
        float x[NMAX];
        float mean = 0.f;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            x[n] = n < N ? p0[(int64_t)n * V * hw + i] : 0.f;
            mean += x[n];
        }
        mean /= (float)N;
        float e = 0.f, sp = 0.f, var = 0.f;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            if (n < N) {
                e += fabsf(x[n] - t);
                var += (x[n] - mean) * (x[n] - mean);
#pragma unroll
                for (int k = 0; k < NMAX; ++k)
                    if (k < n) sp += fabsf(x[n] - x[k]);
            }
        }
        a0 += w * (mean - t) * (mean - t);
        a1 += w * e;
        a2 += w * 2.0f * sp;  // both orders of every pair
        a3 += w * var / (float)(N - 1);
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        red[0][wv] = a0; red[1][wv] = a1; red[2][wv] = a2; red[3][wv] = a3;
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(out + (int64_t)bv * 4 + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] +
                                                                           red[threadIdx.x][2] + red[threadIdx.x][3]);
}


// --------------------------------------------------------------------------------- optimiser step (trainer.py:219-247)
// Gradient sanitising (nan_to_num), Adam / AdamW update and the EMA rule in ONE pass over the parameters: reads g, p, m, v,
// p_ema and writes p, m, v, p_ema (36 B per parameter) where the reference runs nan_to_num, ~10 _foreach passes of
// torch.optim.AdamW and a lerp per tensor.  Work arrives as a chunk table (one block per <= 16384-element chunk of one
// parameter tensor): parameters and EMA copies stay ordinary separately-allocated tensors, gradients and moments are flat.
__global__ __launch_bounds__(256) void adamw_ema_kernel(const swiftk_opt_chunk* __restrict__ chunks, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, swiftk_opt_hyper h) {
    const swiftk_opt_chunk c = chunks[blockIdx.x];
    float* __restrict__ p = c.p;
    float* __restrict__ e = c.ema;
    const float lr = h.lr[c.group], wd = h.weight_decay[c.group], step_size = h.step_size[c.group];
    const float decay = h.decoupled ? 1.0f - lr * wd : 1.0f;
    const float l2 = h.decoupled ? 0.0f : wd;
    const float omb = 1.0f - h.ema_beta;
    for (int i = threadIdx.x; i < c.n; i += 256) {
        float gi = g[c.flat_off + i];
        // torch.nan_to_num(nan=0, posinf=1e5, neginf=-1e5)
        gi = gi != gi ? 0.0f : (gi == INFINITY ? 1e5f : (gi == -INFINITY ? -1e5f : gi));
        g[c.flat_off + i] = gi;
        float pi = p[i] * decay;
        gi += l2 * pi;                                   // Adam (not W): L2 term joins the gradient
        float mi = m[c.flat_off + i], vi = v[c.flat_off + i];
        mi += (gi - mi) * (1.0f - h.beta1);              // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * h.beta2 + (1.0f - h.beta2) * gi * gi;
        const float denom = sqrtf(vi) / h.bias2_sqrt + h.eps;
        pi -= step_size * (mi / denom);
        p[i] = pi;
        m[c.flat_off + i] = mi;
        v[c.flat_off + i] = vi;
        if (e) {                                          // p_ema <- p_net.lerp(p_ema, beta): torch's two-sided lerp formula
            const float ei = e[i];
            e[i] = h.ema_beta < 0.5f ? pi + h.ema_beta * (ei - pi) : ei - (ei - pi) * omb;
        }
    }
}

}  // namespace

int g_modnorm_bwd_fused = 1;  // tuning key 16 (A/B): 0 = row pass and column pass as two kernels

#define DT_SWITCH(dtype, CALL_BF16, CALL_F32) \
    if (dtype == SWIFTK_BF16) { CALL_BF16; } else if (dtype == SWIFTK_F32) { CALL_F32; } else return SWIFTK_EINVAL

extern "C" int swiftk_transpose(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype,
                                void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < rows) return SWIFTK_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((ldd + 63) / 64));
    if (dtype == SWIFTK_BF16 && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15) && !(lds & 7) && !(ldd & 7)) {
        hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, st, static_cast<const bf16_t*>(src), lds,
                           static_cast<bf16_t*>(dst), ldd, rows, cols);
        SWIFTK_CHECK_LAUNCH();
        return 0;
    }
    DT_SWITCH(dtype,
              hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, st, static_cast<const bf16_t*>(src), lds,
                                 static_cast<bf16_t*>(dst), ldd, rows, cols),
              hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(src), lds,
                                 static_cast<float*>(dst), ldd, rows, cols));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_reduce_slabs(const float* slabs, int64_t ld_slab, int64_t slab_stride, int nslabs, float* out,
                                   int64_t ld_out, int64_t rows, int64_t cols, int accumulate, void* stream) {
    if (!slabs || !out || nslabs <= 0 || rows <= 0 || cols <= 0) return SWIFTK_EINVAL;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, static_cast<hipStream_t>(stream), slabs,
                       ld_slab, slab_stride, nslabs, out, ld_out, rows, cols, accumulate);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_swiglu_fwd(const void* h, int64_t ldh, void* out, int64_t ldo, int64_t M, int mlp, int dtype,
                                 void* stream) {
    if (!h || !out || M <= 0 || mlp <= 0) return SWIFTK_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(M * mlp);
    DT_SWITCH(dtype,
              hipLaunchKernelGGL(swiglu_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(h), ldh,
                                 static_cast<bf16_t*>(out), ldo, M, mlp),
              hipLaunchKernelGGL(swiglu_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(h), ldh,
                                 static_cast<float*>(out), ldo, M, mlp));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_swiglu_bwd(const void* h, int64_t ldh, const void* dout, int64_t ldo, void* dh, int64_t lddh, int64_t M,
                                 int mlp, int dtype, void* stream) {
    if (!h || !dout || !dh || M <= 0 || mlp <= 0) return SWIFTK_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(M * mlp);
    DT_SWITCH(dtype,
              hipLaunchKernelGGL(swiglu_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(h), ldh,
                                 static_cast<const bf16_t*>(dout), ldo, static_cast<bf16_t*>(dh), lddh, M, mlp),
              hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(h), ldh,
                                 static_cast<const float*>(dout), ldo, static_cast<float*>(dh), lddh, M, mlp));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

static int modnorm_bwd_impl(const void* y, int64_t ldy, const float* g, void* dy, int64_t lddy, const float* gamma,
                            const float* beta, const float* mod, int64_t ldmod, float* dgamma, float* dbeta, float* dmod,
                            int64_t lddmod, float* row_stats, int64_t M, int d, int64_t rows_per_sample, float eps,
                            int dtype, void* stream, bool ws_zero) {
    if (!y || !g || !dy || !gamma || !beta || !mod || !dgamma || !dbeta || !dmod || !row_stats || M <= 0 || rows_per_sample <= 0)
        return SWIFTK_EINVAL;
    if (d % 4 || d > 1536 || M % rows_per_sample) return SWIFTK_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // one pass (rows and column sums together) when the row-statistics workspace [2 M] can hold the per-sample column sums [2][B][d]
    int rpbf = 64 * g_modnorm_bwd_fused;  // key 16: 1 = the largest of 256 / 128 / 64 rows per block that still gives every CU a block
    if (g_modnorm_bwd_fused == 1) {
        rpbf = 256;
        while (rpbf > 64 && (rows_per_sample % rpbf || M / rpbf < 256)) rpbf >>= 1;
    }
    if (g_modnorm_bwd_fused > 0 && rows_per_sample % rpbf == 0 && rows_per_sample >= d) {
        const unsigned grid = (unsigned)(M / rpbf);
        const int nb = (int)(M / rows_per_sample);
        // psum [2][nb][d] must be zero on entry: cleared here, unless the caller keeps the workspace zero between calls
        // (swiftk_modnorm_bwd_ws0: modnorm_bwd_finish_kernel zeroes what it has read)
        if (!ws_zero && swiftk_zero_f32_impl(row_stats, 2 * (int64_t)nb * d, stream, 1) != 0) return SWIFTK_EINVAL;
        if (!ws_zero && (g_zero_memset & 4)) swiftk_zero_check_launch(row_stats, 2 * (int64_t)nb * d, stream);  // (diagnosis: what did the clear leave?)
#define SWIFTK_MNB(TT, SL)                                                                                                        \
    hipLaunchKernelGGL((modnorm_bwd_fused_kernel<TT, SL>), dim3(grid), dim3(256), 0, st, static_cast<const TT*>(y), ldy, g,       \
                       static_cast<TT*>(dy), lddy, gamma, mod, ldmod, row_stats, M, d, rows_per_sample, eps, rpbf)
        if (dtype == SWIFTK_BF16) {
            if (d <= 1280) SWIFTK_MNB(bf16_t, 5);
            else SWIFTK_MNB(bf16_t, 6);
        } else {
            if (d <= 1280) SWIFTK_MNB(float, 5);
            else SWIFTK_MNB(float, 6);
        }
#undef SWIFTK_MNB
        hipLaunchKernelGGL(modnorm_bwd_finish_kernel, dim3((d + 255) / 256), dim3(256), 0, st, row_stats, gamma, beta, mod, ldmod, dgamma,
                           dbeta, dmod, lddmod, nb, d);
        SWIFTK_CHECK_LAUNCH();
        return 0;
    }
    if (ws_zero) return SWIFTK_ESHAPE;  // (the two-kernel form keeps row statistics in the workspace: it cannot stay zero)
    const unsigned grid_rows = (unsigned)((M + 3) / 4);
    const int rpb = 256;
    const dim3 grid_cols((unsigned)(((d >> 2) + 63) / 64), (unsigned)((M / rows_per_sample) * ((rows_per_sample + rpb - 1) / rpb)));
    DT_SWITCH(dtype,
              {
                  hipLaunchKernelGGL(modnorm_bwd_rows_kernel<bf16_t>, dim3(grid_rows), dim3(256), 0, st,
                                     static_cast<const bf16_t*>(y), ldy, g, static_cast<bf16_t*>(dy), lddy, gamma, mod, ldmod,
                                     row_stats, M, d, rows_per_sample, eps);
                  hipLaunchKernelGGL(modnorm_bwd_cols_kernel<bf16_t>, grid_cols, dim3(256), 0, st, static_cast<const bf16_t*>(y),
                                     ldy, g, gamma, beta, mod, ldmod, row_stats, dgamma, dbeta, dmod, lddmod, d, rows_per_sample,
                                     rpb);
              },
              {
                  hipLaunchKernelGGL(modnorm_bwd_rows_kernel<float>, dim3(grid_rows), dim3(256), 0, st,
                                     static_cast<const float*>(y), ldy, g, static_cast<float*>(dy), lddy, gamma, mod, ldmod,
                                     row_stats, M, d, rows_per_sample, eps);
                  hipLaunchKernelGGL(modnorm_bwd_cols_kernel<float>, grid_cols, dim3(256), 0, st, static_cast<const float*>(y),
                                     ldy, g, gamma, beta, mod, ldmod, row_stats, dgamma, dbeta, dmod, lddmod, d, rows_per_sample,
                                     rpb);
              });
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_qknorm_bwd(const void* qkvh, const void* dqkvh, int64_t ld, const float* rn, void* dqkv, int64_t ldo,
                                 const float* scale, float* dscale, int64_t M, int heads, int head_dim, int dtype,
                                 void* stream) {
    if (!qkvh || !dqkvh || !rn || !dqkv || !scale || !dscale || M <= 0) return SWIFTK_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int inplace = (qkvh != dqkvh && dqkvh == dqkv) ? 1 : 0;  // the gradient buffer then has row stride ldo
    const int grid = grid_for(M * (inplace ? 2 : 3) * heads * 4);  // sixteen lanes per vector, four vectors per group and trip
    DT_SWITCH(dtype,
              hipLaunchKernelGGL(qknorm_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(qkvh),
                                 static_cast<const bf16_t*>(dqkvh), ld, rn, static_cast<bf16_t*>(dqkv), ldo, scale, dscale, M,
                                 heads, head_dim, inplace),
              hipLaunchKernelGGL(qknorm_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(qkvh),
                                 static_cast<const float*>(dqkvh), ld, rn, static_cast<float*>(dqkv), ldo, scale, dscale, M,
                                 heads, head_dim, inplace));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_modnorm_bwd(const void* y, int64_t ldy, const float* g, void* dy, int64_t lddy, const float* gamma,
                                  const float* beta, const float* mod, int64_t ldmod, float* dgamma, float* dbeta, float* dmod,
                                  int64_t lddmod, float* row_stats, int64_t M, int d, int64_t rows_per_sample, float eps,
                                  int dtype, void* stream) {
    return modnorm_bwd_impl(y, ldy, g, dy, lddy, gamma, beta, mod, ldmod, dgamma, dbeta, dmod, lddmod, row_stats, M, d, rows_per_sample,
                            eps, dtype, stream, false);
}

extern "C" int swiftk_modnorm_bwd_ws0(const void* y, int64_t ldy, const float* g, void* dy, int64_t lddy, const float* gamma,
                                      const float* beta, const float* mod, int64_t ldmod, float* dgamma, float* dbeta, float* dmod,
                                      int64_t lddmod, float* workspace, int64_t M, int d, int64_t rows_per_sample, float eps,
                                      int dtype, void* stream) {
    return modnorm_bwd_impl(y, ldy, g, dy, lddy, gamma, beta, mod, ldmod, dgamma, dbeta, dmod, lddmod, workspace, M, d, rows_per_sample,
                            eps, dtype, stream, true);
}

extern "C" int swiftk_colsum(const float* src, int64_t lds, float* out, int64_t rows, int cols, int64_t period, void* stream) {
    if (!src || !out || rows <= 0 || cols <= 0) return SWIFTK_EINVAL;
    const int rpb = 64;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, lds, out, rows, cols, period, rpb);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_embed_bwd_sums(const float* src, int64_t lds, float* bias_grad, float* pos_grad, void* dst_bf16, int64_t ldd,
                                     int64_t rows, int cols, int64_t period, void* stream) {
    if (!src || !bias_grad || !pos_grad || rows <= 0 || cols <= 0 || period <= 0 || lds < cols) return SWIFTK_EINVAL;
    if (rows % period || rows / period > (1 << 20) || (dst_bf16 && (ldd < cols || ldd > (1 << 30)))) return SWIFTK_ESHAPE;
    hipLaunchKernelGGL(embed_bwd_sums_kernel, dim3((unsigned)((period + 7) / 8)), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                       lds, bias_grad, pos_grad, static_cast<bf16_t*>(dst_bf16), ldd, (int)(rows / period), period, cols);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_linear_small_bwd(const float* dz, int64_t lddz, const float* x, int64_t ldx, const float* W, int64_t ldw,
                                       float* dx, int64_t lddx, float* dW, int64_t lddw, float* dbias, int B, int N, int K,
                                       void* stream) {
    if (!dz || !W || B <= 0 || N <= 0 || K <= 0 || B > 64) return SWIFTK_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dx) {  // caller zero-fills dx (partial sums over n-chunks are added atomically)
        const int n_chunk = 128;
        hipLaunchKernelGGL(small_dgrad_kernel, dim3((K + 255) / 256, (N + n_chunk - 1) / n_chunk), dim3(256), 0, st, dz, lddz, W,
                           ldw, dx, lddx, B, N, K, n_chunk);
        SWIFTK_CHECK_LAUNCH();
    }
    if (dW) {
        if (!x) return SWIFTK_EINVAL;
        // rows per block: eight for the big concatenated matrices (block turnover), one or two for a single Linear (a few thousand
        // rows: more blocks, so that every CU has several rows' read-modify-writes in flight)
        if (N >= 16384) hipLaunchKernelGGL(small_wgrad_kernel<8>, dim3((N + 7) / 8), dim3(256), 0, st, dz, lddz, x, ldx, dW, lddw, dbias, B, N, K);
        else if (N >= 4096) hipLaunchKernelGGL(small_wgrad_kernel<2>, dim3((N + 1) / 2), dim3(256), 0, st, dz, lddz, x, ldx, dW, lddw, dbias, B, N, K);
        else hipLaunchKernelGGL(small_wgrad_kernel<1>, dim3(N), dim3(256), 0, st, dz, lddz, x, ldx, dW, lddw, dbias, B, N, K);
        SWIFTK_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int swiftk_silu_bwd(const float* z, const float* dy, float* dz, int64_t n, void* stream) {
    if (!z || !dy || !dz || n <= 0) return SWIFTK_EINVAL;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), z, dy,
                       dz, n);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_crps_loss(const float* preds, const float* target, const float* w_var, const float* w_lat, float* loss,
                                float* dpreds, int m, int B, int C, int H, int W, float alpha, float gscale, void* stream) {
    if (!preds || !target || !w_var || !w_lat || !loss || m < 2 || B <= 0) return SWIFTK_EINVAL;
    const int64_t n = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(crps_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), preds, target, w_var,
                       w_lat, loss, dpreds, m, n, C, H, W, alpha, gscale, 1.0f / ((float)B * H * W));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_trigflow_prep(const float* x, const float* z, const float* t, float* xt_over_sd, float* vt,
                                    float sigma_data, int B, int64_t per_sample, void* stream) {
    if (!x || !z || !t || !xt_over_sd || !vt || B <= 0) return SWIFTK_EINVAL;
    const int64_t n = (int64_t)B * per_sample;
    hipLaunchKernelGGL(trigflow_prep_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, z, t,
                       xt_over_sd, vt, sigma_data, per_sample, n);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_trigflow_loss(const float* F, const float* vt, const float* logvar, const float* w_var,
                                    const float* w_lat, float* loss, float* dF, float* dlogvar, float sigma_data, int B, int C,
                                    int H, int W, float gscale, void* stream) {
    if (!F || !vt || !w_var || !w_lat || !loss || B <= 0) return SWIFTK_EINVAL;
    const int64_t n = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(trigflow_loss_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), F, vt, logvar,
                       w_var, w_lat, loss, dF, dlogvar, sigma_data, n, C, H, W, gscale, 1.0f / ((float)B * H * W));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_axpby_per_sample(float* out, const float* a, const float* x, const float* c, const float* y, int B,
                                       int64_t per_sample, void* stream) {
    if (!out || !a || !x || B <= 0 || (y && !c)) return SWIFTK_EINVAL;
    const int64_t n = (int64_t)B * per_sample;
    hipLaunchKernelGGL(axpby_ps_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), out, a, x, c, y,
                       per_sample, n);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_channel_axpy(float* out, const float* x, const float* y, const float* coef, int B, int C, int64_t hw,
                                   void* stream) {
    if (!out || !y || !coef || B <= 0 || C <= 0) return SWIFTK_EINVAL;
    const int64_t n = (int64_t)B * C * hw;
    hipLaunchKernelGGL(chan_axpy_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), out, x, y, coef, C,
                       hw, n);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_rmse_sums(const float* y, const float* t, int64_t t_batch_stride, const float* w_lat, float* sq, int B,
                                int C, int H, int W, void* stream) {
    if (!y || !t || !w_lat || !sq || B <= 0 || C <= 0 || H <= 0 || W <= 0) return SWIFTK_EINVAL;
    const dim3 grid((unsigned)grid_for((int64_t)B * H * W, 256, 64), (unsigned)C);
    hipLaunchKernelGGL(rmse_sums_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), y, t, t_batch_stride, w_lat, sq, B,
                       C, H, W);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_ensemble_sums(const float* pred, const float* y, const float* w_lat, float* out, int B, int N, int V, int H,
                                    int W, void* stream) {
    if (!pred || !y || !w_lat || !out || B <= 0 || N < 2 || V <= 0 || H <= 0 || W <= 0) return SWIFTK_EINVAL;
    if (N > 64) return SWIFTK_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)grid_for((int64_t)H * W, 256, 32), (unsigned)(B * V));
    if (N <= 8)
        hipLaunchKernelGGL(ensemble_sums_kernel<8>, grid, dim3(256), 0, st, pred, y, w_lat, out, N, V, H, W);
    else if (N <= 16)
        hipLaunchKernelGGL(ensemble_sums_kernel<16>, grid, dim3(256), 0, st, pred, y, w_lat, out, N, V, H, W);
    else
        hipLaunchKernelGGL(ensemble_sums_kernel<64>, grid, dim3(256), 0, st, pred, y, w_lat, out, N, V, H, W);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_adamw_ema_step(const swiftk_opt_chunk* chunks, int n_chunks, float* grad_flat, float* exp_avg_flat,
                                     float* exp_avg_sq_flat, const swiftk_opt_hyper* hyper_host, void* stream) {
    if (!chunks || !grad_flat || !exp_avg_flat || !exp_avg_sq_flat || !hyper_host || n_chunks <= 0) return SWIFTK_EINVAL;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream), chunks, grad_flat,
                       exp_avg_flat, exp_avg_sq_flat, *hyper_host);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}
