// Forward-mode (tangent) kernels of the sCM pre-training loss for gfx950 (reference src/swift/training/loss.py:186-260:
// torch.func.jvp through the denoiser with jvp=True, i.e. the explicit softmax(q k^T) v attention of swinv2.py:129-133).
//
// The network's linear maps carry the tangent as extra rows of the same GEMM (primal rows 0..M-1, tangent rows M..2M-1,
// gemm.hip unchanged); what lives here are the tangent rules of the non-linear steps:
//   time embedding, SiLU, q/k L2-normalisation, windowed softmax attention, LayerNorm + modulation, SwiGLU,
// and the loss-side target construction.  Element-wise arithmetic and statistics are fp32 (bf16 only as storage).  The
// attention products run on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32) for fp32 operands -- the parity configuration --
// and on the bf16 pipe (v_mfma_f32_16x16x32_bf16) for bf16 operands, as the reference's autocast does.
#include "common.h"

namespace {

inline int grid_for(int64_t work_items, int per_block = 256, int cap = 256 * 16) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return elem<T>::to_f(*p); }

constexpr float LN100 = 4.605170185988092f;

// --------------------------------------------------------------------------------- time embedding / SiLU tangents
// emb = [sin(t w f) | cos(t w f)] (swinv2.py:44-60)  =>  d emb = [cos(.) | -sin(.)] * w f * dt
__global__ void temb_jvp_kernel(const float* __restrict__ t, const float* __restrict__ dt, const float* __restrict__ freqs,
                                float* __restrict__ demb, int B, int d, float tw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d) return;
    const int b = i / d, k = i - b * d, half = d >> 1;
    float v = 0.f;
    if (k < 2 * half) {
        const float f = freqs[k < half ? k : k - half];
        const float arg = (t[b] * tw) * f;
        v = (k < half ? cosf(arg) : -sinf(arg)) * (tw * f) * dt[b];
    }
    demb[i] = v;
}

__global__ void silu_jvp_kernel(const float* __restrict__ z, const float* __restrict__ dz, float* __restrict__ y,
                                float* __restrict__ dy, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = z[i], s = 1.0f / (1.0f + expf(-v));
        if (y) y[i] = v * s;
        dy[i] = dz[i] * (s + v * s * (1.0f - s));
    }
}

// --------------------------------------------------------------------------------- q/k normalisation tangent
// v_hat = v / max(|v|, 1e-12) * tau,  d v_hat = tau / n * (dv - v (v . dv) / n^2)   (swinv2.py:123-127; tau = 1 for k)
// One 16-lane group per (row, head, q|k) vector of head_dim (80 / 88 / 96); in place on the primal and the tangent tensor.
template <typename T, int HD>
__global__ __launch_bounds__(256) void qknorm_jvp_kernel(T* __restrict__ qkv, T* __restrict__ dqkv, int64_t ld,
                                                         const float* __restrict__ scale, float* __restrict__ rn, int64_t M,
                                                         int heads) {
    const int l16 = threadIdx.x & 15;
    const int64_t nvec = M * heads * 2;
    for (int64_t vid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; vid < nvec; vid += ((int64_t)gridDim.x * 256) >> 4) {
        const int part = (int)(vid & 1);
        const int64_t mh = vid >> 1;
        const int h = (int)(mh % heads);
        const int64_t m = mh / heads;
        T* p = qkv + m * ld + (h * 3 + part) * HD;
        T* dp = dqkv + m * ld + (h * 3 + part) * HD;
        float v[6], dv[6], ss = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = l16 + 16 * i;
            v[i] = e < HD ? ldf(p + e) : 0.f;
            dv[i] = e < HD ? ldf(dp + e) : 0.f;
            ss += v[i] * v[i];
            dot += v[i] * dv[i];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            ss += __shfl_xor(ss, o, 64);
            dot += __shfl_xor(dot, o, 64);
        }
        const float n = fmaxf(sqrtf(ss), 1e-12f);
        const float tau = part == 0 ? expf(fminf(scale[h], LN100)) : 1.0f;
        const float a = tau / n, c = dot / (n * n);
        if (rn && l16 == 0) {  // what SWIFTK_EPI_QKNORM saves for the backward pass: 1 / max(|.|, 1e-12) per q / k vector, 1 for v
            rn[m * (3 * heads) + h * 3 + part] = 1.0f / n;
            if (part == 0) rn[m * (3 * heads) + h * 3 + 2] = 1.0f;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = l16 + 16 * i;
            if (e < HD) {
                p[e] = elem<T>::from_f(v[i] * a);
                dp[e] = elem<T>::from_f(a * (dv[i] - v[i] * c));
            }
        }
    }
}

// --------------------------------------------------------------------------------- SwiGLU tangent
// h [M, 2*mlp] with columns interleaved (gate_j, up_j):  o = silu(g) u,  do = silu'(g) dg u + silu(g) du
template <typename T>
__global__ __launch_bounds__(256) void swiglu_jvp_kernel(const T* __restrict__ h, const T* __restrict__ dh, int64_t ldh,
                                                         T* __restrict__ o, T* __restrict__ dout, int64_t ldo, int64_t M,
                                                         int mlp) {
    const int64_t total = M * mlp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / mlp;
        const int j = (int)(i - m * mlp);
        const float g = ldf(h + m * ldh + 2 * j), u = ldf(h + m * ldh + 2 * j + 1);
        const float dg = ldf(dh + m * ldh + 2 * j), du = ldf(dh + m * ldh + 2 * j + 1);
        const float s = 1.0f / (1.0f + expf(-g));
        o[m * ldo + j] = elem<T>::from_f(g * s * u);
        dout[m * ldo + j] = elem<T>::from_f((s + g * s * (1.0f - s)) * dg * u + g * s * du);
    }
}

// --------------------------------------------------------------------------------- LayerNorm + modulation tangent
// primal:  x += (n gamma + beta)(1 + sc) + sh,          n = (y - mu) rstd            (swinv2.py:77-86, post-norm :137,:101)
// tangent: dx += gamma dn (1 + sc) + (n gamma + beta) dsc + dsh,  dn = (dy - mean(dy) - n mean(n dy)) rstd
// One wave per row; the row (d <= 1536) lives in registers.  Scalar form for rows that are not whole 8-channel chunks.
template <typename T>
__global__ __launch_bounds__(256) void modnorm_jvp_scalar_kernel(const T* __restrict__ y, const T* __restrict__ dy, int64_t ldy,
                                                          float* __restrict__ x, float* __restrict__ dx, T* __restrict__ xT,
                                                          T* __restrict__ dxT, int64_t ldxT, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ mod,
                                                          const float* __restrict__ dmod, int64_t ldmod, int64_t M, int d,
                                                          int64_t rps, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t b = row / rps;
    float yv[24], dv[24];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const int e = lane + 64 * i;
        yv[i] = e < d ? ldf(y + row * ldy + e) : 0.f;
        dv[i] = e < d ? ldf(dy + row * ldy + e) : 0.f;
        s0 += yv[i];
        s1 += dv[i];
    }
    const float inv_d = 1.0f / (float)d;
    const float mu = wave_sum(s0) * inv_d, mdy = wave_sum(s1) * inv_d;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const int e = lane + 64 * i;
        const float c = e < d ? yv[i] - mu : 0.f;
        s2 += c * c;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) * inv_d + eps);
    float s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const int e = lane + 64 * i;
        yv[i] = e < d ? (yv[i] - mu) * rstd : 0.f;  // n
        s3 += yv[i] * dv[i];
    }
    const float mndy = wave_sum(s3) * inv_d;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const int e = lane + 64 * i;
        if (e < d) {
            const float n = yv[i], dn = (dv[i] - mdy - n * mndy) * rstd;
            const float ga = gamma[e], ln = n * ga + beta[e];
            const float sc = mod[b * ldmod + e], sh = mod[b * ldmod + d + e];
            const float dsc = dmod[b * ldmod + e], dsh = dmod[b * ldmod + d + e];
            const float xn = x[row * d + e] + ln * (1.0f + sc) + sh;
            const float dxn = dx[row * d + e] + ga * dn * (1.0f + sc) + ln * dsc + dsh;
            x[row * d + e] = xn;
            dx[row * d + e] = dxn;
            xT[row * ldxT + e] = elem<T>::from_f(xn);
            dxT[row * ldxT + e] = elem<T>::from_f(dxn);
        }
    }
}

// Vector form: 8 channels per lane and slot (16-B bf16 / 32-B fp32 accesses like modnorm_kernel); the scalar form above
// moves 2-4 B per access and runs at 3.1 TB/s, this one at the ModulatedNorm kernel's rate.
template <typename T>
__global__ __launch_bounds__(256) void modnorm_jvp_kernel(const T* __restrict__ y, const T* __restrict__ dy, int64_t ldy,
                                                          float* __restrict__ x, float* __restrict__ dx, T* __restrict__ xT,
                                                          T* __restrict__ dxT, int64_t ldxT, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ mod,
                                                          const float* __restrict__ dmod, int64_t ldmod, int64_t M, int d,
                                                          int64_t rps, float eps) {
    constexpr int SLOTS = 3;  // d <= 1536
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t b = row / rps;
    const int nc = d >> 3;
    float yv[SLOTS][8], dv[SLOTS][8];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            load8<T>(y + row * ldy + 8 * c, yv[i]);
            load8<T>(dy + row * ldy + 8 * c, dv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s0 += yv[i][e];
                s1 += dv[i][e];
            }
        }
    }
    const float inv_d = 1.0f / (float)d;
    const float mu = wave_sum(s0) * inv_d, mdy = wave_sum(s1) * inv_d;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
        if (lane + 64 * i < nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                yv[i][e] -= mu;
                s2 += yv[i][e] * yv[i][e];
            }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) * inv_d + eps);
    float s3 = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
        if (lane + 64 * i < nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                yv[i][e] *= rstd;  // n
                s3 += yv[i][e] * dv[i][e];
            }
        }
    const float mndy = wave_sum(s3) * inv_d;
    const float* mrow = mod + b * ldmod;
    const float* dmrow = dmod + b * ldmod;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            float ga[8], be[8], sc[8], sh[8], dsc[8], dsh[8], xr[8], dxr[8];
            load8<float>(gamma + 8 * c, ga);
            load8<float>(beta + 8 * c, be);
            load8<float>(mrow + 8 * c, sc);
            load8<float>(mrow + d + 8 * c, sh);
            load8<float>(dmrow + 8 * c, dsc);
            load8<float>(dmrow + d + 8 * c, dsh);
            load8<float>(x + row * d + 8 * c, xr);
            load8<float>(dx + row * d + 8 * c, dxr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = yv[i][e], dn = (dv[i][e] - mdy - n * mndy) * rstd;
                const float ln = n * ga[e] + be[e];
                xr[e] += ln * (1.0f + sc[e]) + sh[e];
                dxr[e] += ga[e] * dn * (1.0f + sc[e]) + ln * dsc[e] + dsh[e];
            }
            store8<float>(x + row * d + 8 * c, xr);
            store8<float>(dx + row * d + 8 * c, dxr);
            store8<T>(xT + row * ldxT + 8 * c, xr);
            store8<T>(dxT + row * ldxT + 8 * c, dxr);
        }
    }
}

// Pair form (bf16 operands): the residual stream and its tangent are held as (bf16 hi, 8-bit lo) pairs -- hi is the GEMM operand
// (xT / dxT), lo one byte in units of ulp(hi) / 256 (common.h) -- instead of fp32 x, dx plus operand copies: 16 bytes per element
// (y, dy 4; hi 4 + lo 2 in; hi 4 + lo 2 out) where the fp32 form moves 24.  hi is read from (xT_in, dxT_in) and written to
// (xT, dxT): the same buffers when the pass keeps nothing, the previous layer's saved operand otherwise.
__global__ __launch_bounds__(256) void modnorm_jvp_pair_kernel(const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy, int64_t ldy,
                                                               const bf16_t* xT_in, const bf16_t* dxT_in, bf16_t* xT, bf16_t* dxT,
                                                               int64_t ldxT, uint8_t* __restrict__ xlo, uint8_t* __restrict__ dxlo,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ mod, const float* __restrict__ dmod,
                                                               int64_t ldmod, int64_t M, int d, int64_t rps, float eps) {
    constexpr int SLOTS = 3;  // d <= 1536
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t b = row / rps;
    const int nc = d >> 3;
    float yv[SLOTS][8], dv[SLOTS][8];
    uint4 hx[SLOTS], hd[SLOTS];
    uint2 lx[SLOTS], ld_[SLOTS];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            load8<bf16_t>(y + row * ldy + 8 * c, yv[i]);
            load8<bf16_t>(dy + row * ldy + 8 * c, dv[i]);
            hx[i] = *reinterpret_cast<const uint4*>(xT_in + row * ldxT + 8 * c);
            hd[i] = *reinterpret_cast<const uint4*>(dxT_in + row * ldxT + 8 * c);
            lx[i] = *reinterpret_cast<const uint2*>(xlo + row * d + 8 * c);
            ld_[i] = *reinterpret_cast<const uint2*>(dxlo + row * d + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s0 += yv[i][e];
                s1 += dv[i][e];
            }
        }
    }
    const float inv_d = 1.0f / (float)d;
    const float mu = wave_sum(s0) * inv_d, mdy = wave_sum(s1) * inv_d;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
        if (lane + 64 * i < nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                yv[i][e] -= mu;
                s2 += yv[i][e] * yv[i][e];
            }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) * inv_d + eps);
    float s3 = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
        if (lane + 64 * i < nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                yv[i][e] *= rstd;  // n
                s3 += yv[i][e] * dv[i][e];
            }
        }
    const float mndy = wave_sum(s3) * inv_d;
    const float* mrow = mod + b * ldmod;
    const float* dmrow = dmod + b * ldmod;
    auto value = [](const uint4& h, const uint2& l, float (&v)[8]) {  // the eight fp32 values a (hi, lo) slot stands for
        const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[2] = {l.x, l.y};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float hf = (e & 1) ? __uint_as_float(hw[e >> 1] & 0xffff0000u) : __uint_as_float(hw[e >> 1] << 16);
            const uint32_t E = (hw[e >> 1] >> ((e & 1) ? 23 : 7)) & 0xFFu;
            v[e] = hf + lo8_value((float)((lw[e >> 2] >> (8 * (e & 3))) & 0xFFu), E);
        }
    };
    auto store_pair = [](const float (&v)[8], bf16_t* hp, uint8_t* lp) {
        uint32_t oh[4], ol[2] = {0u, 0u};
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const uint32_t ph = pack_bf16(v[2 * e2], v[2 * e2 + 1]);
            oh[e2] = ph;
            ol[e2 >> 1] = lo8_insert(v[2 * e2], __uint_as_float(ph << 16), (ph >> 7) & 0xFFu, (2 * e2) & 3, ol[e2 >> 1]);
            ol[e2 >> 1] = lo8_insert(v[2 * e2 + 1], __uint_as_float(ph & 0xffff0000u), (ph >> 23) & 0xFFu, (2 * e2 + 1) & 3, ol[e2 >> 1]);
        }
        *reinterpret_cast<uint4*>(hp) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
        *reinterpret_cast<uint2*>(lp) = make_uint2(ol[0], ol[1]);
    };
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            float ga[8], be[8], sc[8], sh[8], dsc[8], dsh[8], xr[8], dxr[8];
            load8<float>(gamma + 8 * c, ga);
            load8<float>(beta + 8 * c, be);
            load8<float>(mrow + 8 * c, sc);
            load8<float>(mrow + d + 8 * c, sh);
            load8<float>(dmrow + 8 * c, dsc);
            load8<float>(dmrow + d + 8 * c, dsh);
            value(hx[i], lx[i], xr);
            value(hd[i], ld_[i], dxr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = yv[i][e], dn = (dv[i][e] - mdy - n * mndy) * rstd;
                const float ln = n * ga[e] + be[e];
                xr[e] += ln * (1.0f + sc[e]) + sh[e];
                dxr[e] += ga[e] * dn * (1.0f + sc[e]) + ln * dsc[e] + dsh[e];
            }
            store_pair(xr, xT + row * ldxT + 8 * c, xlo + row * d + 8 * c);
            store_pair(dxr, dxT + row * ldxT + 8 * c, dxlo + row * d + 8 * c);
        }
    }
}

// The same, a block walking `rows_per_block` rows of one sample (round 4): the six per-column constant vectors enter only through
//     A = gamma (1+sc),  B = beta (1+sc) + sh,  C = gamma dsc,  D = beta dsc + dsh:   x += n A + B,   dx += dn A + n C + D,
// formed once per block in LDS (the row-per-wave kernel above pulls 6 x 4 KB through L2 for every row, after its reductions); the
// next row's loads are in flight while the current row is reduced and stored, and the row's four sums -- with t = y - y0:
// sum t, sum t^2, sum dy, sum t dy, from which mean, rstd, mean(dy), mean(n dy) follow -- cross the wave in ONE round of adds
// instead of three dependent ones.
__global__ __launch_bounds__(256) void modnorm_jvp_pair_rows_kernel(const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy, int64_t ldy,
                                                                    const bf16_t* xT_in, const bf16_t* dxT_in, bf16_t* xT, bf16_t* dxT,
                                                                    int64_t ldxT, uint8_t* xlo, uint8_t* dxlo,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ mod, const float* __restrict__ dmod,
                                                                    int64_t ldmod, int64_t M, int d, int64_t rps, float eps,
                                                                    int rows_per_block) {
    constexpr int SLOTS = 3, DP = SLOTS * 512;  // d <= 1536
    __shared__ __attribute__((aligned(16))) float cst[4][DP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, M);
    const int64_t b = r0 / rps;
    const int nc = d >> 3;
    struct Raw {
        uint4 y, dy, hx, hd;
        uint2 lx, ld;
    };
    Raw nx[SLOTS];
    auto fetch = [&](int64_t row) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                nx[i].y = *reinterpret_cast<const uint4*>(y + row * ldy + 8 * c);
                nx[i].dy = *reinterpret_cast<const uint4*>(dy + row * ldy + 8 * c);
                nx[i].hx = *reinterpret_cast<const uint4*>(xT_in + row * ldxT + 8 * c);
                nx[i].hd = *reinterpret_cast<const uint4*>(dxT_in + row * ldxT + 8 * c);
                nx[i].lx = *reinterpret_cast<const uint2*>(xlo + row * d + 8 * c);
                nx[i].ld = *reinterpret_cast<const uint2*>(dxlo + row * d + 8 * c);
            }
        }
    };
    if (r0 + wv < r1) fetch(r0 + wv);
    {
        const float* mrow = mod + b * ldmod;
        const float* dmrow = dmod + b * ldmod;
        for (int col = threadIdx.x; col < DP; col += 256) {
            float A = 0.f, Bc = 0.f, C = 0.f, D = 0.f;
            if (col < d) {
                const float ga = gamma[col], be = beta[col], sc1 = 1.0f + mrow[col], dsc = dmrow[col];
                A = ga * sc1;
                Bc = be * sc1 + mrow[d + col];
                C = ga * dsc;
                D = be * dsc + dmrow[d + col];
            }
            cst[0][col] = A; cst[1][col] = Bc; cst[2][col] = C; cst[3][col] = D;
        }
    }
    __syncthreads();
    auto unpack8 = [](const uint4& u, float (&v)[8]) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __uint_as_float(w[e] << 16);
            v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        }
    };
    auto value = [](const uint4& h, const uint2& l, float (&v)[8]) {  // the eight fp32 values a (hi, lo) slot stands for
        const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[2] = {l.x, l.y};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float hf = (e & 1) ? __uint_as_float(hw[e >> 1] & 0xffff0000u) : __uint_as_float(hw[e >> 1] << 16);
            const uint32_t E = (hw[e >> 1] >> ((e & 1) ? 23 : 7)) & 0xFFu;
            v[e] = hf + lo8_value((float)((lw[e >> 2] >> (8 * (e & 3))) & 0xFFu), E);
        }
    };
    auto store_pair = [](const float (&v)[8], bf16_t* hp, uint8_t* lp) {
        uint32_t oh[4], ol[2] = {0u, 0u};
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const uint32_t ph = pack_bf16(v[2 * e2], v[2 * e2 + 1]);
            oh[e2] = ph;
            ol[e2 >> 1] = lo8_insert(v[2 * e2], __uint_as_float(ph << 16), (ph >> 7) & 0xFFu, (2 * e2) & 3, ol[e2 >> 1]);
            ol[e2 >> 1] = lo8_insert(v[2 * e2 + 1], __uint_as_float(ph & 0xffff0000u), (ph >> 23) & 0xFFu, (2 * e2 + 1) & 3, ol[e2 >> 1]);
        }
        *reinterpret_cast<uint4*>(hp) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
        *reinterpret_cast<uint2*>(lp) = make_uint2(ol[0], ol[1]);
    };
    const float inv_d = 1.0f / (float)d;
    for (int64_t row = r0 + wv; row < r1; row += 4) {
        Raw cur[SLOTS];
        float yv[SLOTS][8], dv[SLOTS][8];
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            cur[i] = nx[i];
            if (lane + 64 * i < nc) {
                unpack8(cur[i].y, yv[i]);
                unpack8(cur[i].dy, dv[i]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) yv[i][e] = dv[i][e] = 0.f;
            }
        }
        if (row + 4 < r1) fetch(row + 4);
        const float y0 = __shfl(yv[0][0], 0, 64);
        float q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = (lane + 64 * i < nc) ? yv[i][e] - y0 : 0.f;
                yv[i][e] = t;
                q1 += t;
                q2 = fmaf(t, t, q2);
                q3 += dv[i][e];
                q4 = fmaf(t, dv[i][e], q4);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            q1 += __shfl_xor(q1, o, 64);
            q2 += __shfl_xor(q2, o, 64);
            q3 += __shfl_xor(q3, o, 64);
            q4 += __shfl_xor(q4, o, 64);
        }
        const float mt = q1 * inv_d;
        const float rstd = 1.0f / sqrtf(fmaxf(q2 * inv_d - mt * mt, 0.f) + eps);
        const float mdy = q3 * inv_d, mndy = rstd * (q4 - mt * q3) * inv_d;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = lane + 64 * i;
            if (c < nc) {
                float xr[8], dxr[8];
                value(cur[i].hx, cur[i].lx, xr);
                value(cur[i].hd, cur[i].ld, dxr);
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const float4 a4 = *reinterpret_cast<const float4*>(&cst[0][8 * c + 4 * hq]);
                    const float4 b4 = *reinterpret_cast<const float4*>(&cst[1][8 * c + 4 * hq]);
                    const float4 c4 = *reinterpret_cast<const float4*>(&cst[2][8 * c + 4 * hq]);
                    const float4 d4 = *reinterpret_cast<const float4*>(&cst[3][8 * c + 4 * hq]);
                    const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
                    const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int e = 4 * hq + e4;
                        const float n = (yv[i][e] - mt) * rstd, dn = (dv[i][e] - mdy - n * mndy) * rstd;
                        xr[e] += fmaf(n, aa[e4], bb[e4]);
                        dxr[e] += fmaf(dn, aa[e4], fmaf(n, cc[e4], dd[e4]));
                    }
                }
                store_pair(xr, xT + row * ldxT + 8 * c, xlo + row * d + 8 * c);
                store_pair(dxr, dxT + row * ldxT + 8 * c, dxlo + row * d + 8 * c);
            }
        }
    }
}

// --------------------------------------------------------------------------------- windowed softmax attention tangent
// Per (sample, window, head), q_hat/k_hat pre-normalised (qknorm_jvp), softmax scale 1 (swinv2.py:129-133):
//   S = Q K^T,  P = softmax(S),  O = P V
//   dS = dQ K^T + Q dK^T,  dP = P o (dS - rowsum(P o dS)),  dO = dP V + P dV
// With unnormalised e = exp(S - max), l = rowsum(e), W = e o dS, r = rowsum(W):
//   O = (e V) / l,   dO = (W V + e dV - (r / l) (e V)) / l
// One workgroup per (item, query half): 8 waves x 16 query rows, v_mfma_f32_16x16x4_f32.  One fp32 LDS image
// (256 rows x 96, stride 100 floats: conflict-free for both read patterns below) holds K, dK, V, dV in turn.
// S^T[key][q] orientation: a lane owns query column c16 and keys 16 blk + 4 g + r (g = lane >> 4), so the accumulator
// registers are, register for register, the B operand of the four-key k-steps {16 blk + 4 kk + r : kk = 0..3} of
// O^T[d][q] = V^T[d][key] P^T[key][q].
constexpr int ISTR = 100;

struct AttnJvpArgs {
    const void* qkv;
    const void* dqkv;
    void* out;
    void* dout;
    int64_t ldq, ldo;
    int gh, gw, heads, sh, sw, nwx, nw;
};

__device__ __forceinline__ int jvp_window_token(const AttnJvpArgs& a, int w, int j) {
    const int wy = w / a.nwx, wx = w - wy * a.nwx;
    int gy = wy * 16 + (j >> 4) + a.sh;
    int gx = wx * 16 + (j & 15) + a.sw;
    gy = gy >= a.gh ? gy - a.gh : gy;
    gx = gx >= a.gw ? gx - a.gw : gx;
    return gy * a.gw + gx;
}

template <typename T, int HD>
struct RowLoad;
template <int HD>
struct RowLoad<float, HD> {
    __device__ static __forceinline__ void run(const float* src, float* dst) {
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) *reinterpret_cast<float4*>(dst + 4 * c) = *reinterpret_cast<const float4*>(src + 4 * c);
    }
};
template <int HD>
struct RowLoad<bf16_t, HD> {
    __device__ static __forceinline__ void run(const bf16_t* src, float* dst) {
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(src + 8 * c);
            *reinterpret_cast<float4*>(dst + 8 * c) = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                                                  __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
            *reinterpret_cast<float4*>(dst + 8 * c + 4) = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                                                                      __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
        }
    }
};

template <typename T, int HD>
__global__ __launch_bounds__(512) void attn_jvp_kernel(AttnJvpArgs a) {
    __shared__ __attribute__((aligned(16))) float img[256 * ISTR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qh = blockIdx.x & 1;
    const int item = blockIdx.x >> 1;
    const int head = item % a.heads;
    const int w = (item / a.heads) % a.nw;
    const int b = item / (a.heads * a.nw);
    const int64_t tok0 = (int64_t)b * a.gh * a.gw;
    const T* qkv = static_cast<const T*>(a.qkv);
    const T* dqkv = static_cast<const T*>(a.dqkv);
    const int c16 = lane & 15, g = lane >> 4;

    // fill the image with part `part` (1 = k, 2 = v) of the primal (tan = 0) or tangent (tan = 1) tensor
    auto fill = [&](int part, int tan) {
        if (tid < 256) {
            const T* src = (tan ? dqkv : qkv) + (tok0 + jvp_window_token(a, w, tid)) * a.ldq + (head * 3 + part) * HD;
            float* dst = img + tid * ISTR;
            RowLoad<T, HD>::run(src, dst);
#pragma unroll
            for (int z = HD; z < 96; z += 4) *reinterpret_cast<float4*>(dst + z) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    // this lane's slice of its query row: d = 24 g + i, i = 0..23 (zero for d >= head_dim)
    float qv[24], dqv[24];
    {
        const int tq = jvp_window_token(a, w, qh * 128 + wv * 16 + c16);
        const T* qs = qkv + (tok0 + tq) * a.ldq + head * 3 * HD;
        const T* dqs = dqkv + (tok0 + tq) * a.ldq + head * 3 * HD;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const int d = 24 * g + i;
            qv[i] = d < HD ? ldf(qs + d) : 0.f;
            dqv[i] = d < HD ? ldf(dqs + d) : 0.f;
        }
    }
    fill(1, 0);
    __syncthreads();

    f32x4 s[16], ds[16];
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        s[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
        ds[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* krow = img + (blk * 16 + c16) * ISTR + 24 * g;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(krow + 4 * c);
            const float kk[4] = {kf.x, kf.y, kf.z, kf.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[e], qv[4 * c + e], s[blk], 0, 0, 0);
                ds[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[e], dqv[4 * c + e], ds[blk], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    fill(1, 1);
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        const float* krow = img + (blk * 16 + c16) * ISTR + 24 * g;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(krow + 4 * c);
            const float kk[4] = {kf.x, kf.y, kf.z, kf.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) ds[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[e], qv[4 * c + e], ds[blk], 0, 0, 0);
        }
    }

    // softmax statistics of query column c16: 64 keys here, the rest in lanes c16 + 16, + 32, + 48
    float mx = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < 16; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[blk][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f, rs = 0.f;
#pragma unroll
    for (int blk = 0; blk < 16; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = expf(s[blk][r] - mx);
            const float wgt = e * ds[blk][r];
            s[blk][r] = e;
            ds[blk][r] = wgt;
            l += e;
            rs += wgt;
        }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);

    __syncthreads();
    fill(2, 0);
    __syncthreads();
    f32x4 o[6], u[6], tt[6];
#pragma unroll
    for (int db = 0; db < 6; ++db) {
        o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
        u[db] = f32x4{0.f, 0.f, 0.f, 0.f};
        tt[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int blk = 0; blk < 16; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vrow = img + (blk * 16 + 4 * g + r) * ISTR + c16;
#pragma unroll
            for (int db = 0; db < 6; ++db) {
                const float vf = vrow[16 * db];
                o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, s[blk][r], o[db], 0, 0, 0);
                u[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, ds[blk][r], u[db], 0, 0, 0);
            }
        }
    __syncthreads();
    fill(2, 1);
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < 16; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vrow = img + (blk * 16 + 4 * g + r) * ISTR + c16;
#pragma unroll
            for (int db = 0; db < 6; ++db) tt[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16 * db], s[blk][r], tt[db], 0, 0, 0);
        }

    // lane holds O^T[d = 16 db + 4 g + 0..3][q = c16]
    const float rl = 1.0f / l, rr = rs * rl;
    const int64_t orow = (tok0 + jvp_window_token(a, w, qh * 128 + wv * 16 + c16)) * a.ldo + head * HD;
    T* po = static_cast<T*>(a.out) + orow;
    T* pdo = static_cast<T*>(a.dout) + orow;
#pragma unroll
    for (int db = 0; db < 6; ++db) {
        const int d = 16 * db + 4 * g;
        if (d < HD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                po[d + r] = elem<T>::from_f(o[db][r] * rl);
                pdo[d + r] = elem<T>::from_f((u[db][r] + tt[db][r] - rr * o[db][r]) * rl);
            }
        }
    }
}

// bf16 operands: the same tangent on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulation) -- what the
// reference computes under the trainer's autocast, where q k^T, dq k^T + q dk^T, P v, dP v and P dv are bf16 matmuls.
// K / dK images (208-B rows) are resident together for the score products, then V / dV images (192-B rows) for the value
// products; V^T fragments come from the row-major images through ds_read_b64_tr_b16, with the k-slots of a 32-key step
// ordered to match the S^T accumulator registers of two 16-key blocks: slot (g, j) <-> key 16 (2p + j/4) + 4 g + j%4.
// Round 4: the images are staged in HALVES of 128 keys (53 KB instead of 106: two workgroups per CU, one computing while the other
// waits), and the next half's 16-B chunks are requested into registers before the current half is consumed -- the kernel used to
// spend 32 us per workgroup on 2 us of matrix work, every image load exposed.  The two query halves of an item run on workgroups
// 8 apart in the launch order, i.e. on the same XCD: the second one finds K, dK, V, dV in that XCD's L2.
constexpr int JK = 208, JV = 192;

template <int HD>
__global__ __launch_bounds__(512) void attn_jvp_bf16_kernel(AttnJvpArgs a) {
    __shared__ __attribute__((aligned(16))) char img[2 * 128 * JK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qh = blockIdx.x & 1, item = blockIdx.x >> 1;
    if ((gridDim.x & 15) == 0) {  // (item, query half) = (16 j + i, h) for block 16 j + 8 h + i
        qh = (blockIdx.x >> 3) & 1;
        item = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    }
    const int head = item % a.heads;
    const int w = (item / a.heads) % a.nw;
    const int b = item / (a.heads * a.nw);
    const int64_t tok0 = (int64_t)b * a.gh * a.gw;
    const bf16_t* qkv = static_cast<const bf16_t*>(a.qkv);
    const bf16_t* dqkv = static_cast<const bf16_t*>(a.dqkv);
    const int l16 = lane & 15, g = lane >> 4;

    // one half image pair of `part` (1 = k, 2 = v): 128 keys x 12 chunks, primal rows 0..127, tangent rows 128..255 of img;
    // chunk c = tid + 512 k of [tangent][key][chunk]: six per thread
    uint4 pre[6];
    auto request = [&](int part, int half) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int c = tid + 512 * k;
            const int tan = c >= 128 * 12;
            const int cc0 = c - tan * 128 * 12;
            const int row = cc0 / 12, cc = cc0 - row * 12;
            pre[k] = make_uint4(0, 0, 0, 0);
            if (cc < HD / 8) {
                const bf16_t* src = (tan ? dqkv : qkv) + (tok0 + jvp_window_token(a, w, half * 128 + row)) * a.ldq + (head * 3 + part) * HD;
                pre[k] = *reinterpret_cast<const uint4*>(src + 8 * cc);
            }
        }
    };
    auto deposit = [&](int stride) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int c = tid + 512 * k;
            const int tan = c >= 128 * 12;
            const int cc0 = c - tan * 128 * 12;
            const int row = cc0 / 12, cc = cc0 - row * 12;
            *reinterpret_cast<uint4*>(img + (tan * 128 + row) * stride + cc * 16) = pre[k];
        }
    };

    request(1, 0);
    // q / dq fragments of query row qh*128 + wv*16 + l16: 8 bf16 at d = 32 ks + 8 g (zero beyond head_dim)
    uint4 qf[3], dqf[3];
    {
        const int64_t r = (tok0 + jvp_window_token(a, w, qh * 128 + wv * 16 + l16)) * a.ldq + head * 3 * HD;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int d = 32 * ks + 8 * g;
            qf[ks] = d < HD ? *reinterpret_cast<const uint4*>(qkv + r + d) : make_uint4(0, 0, 0, 0);
            dqf[ks] = d < HD ? *reinterpret_cast<const uint4*>(dqkv + r + d) : make_uint4(0, 0, 0, 0);
        }
    }
    deposit(JK);
    __syncthreads();

    f32x4 s[16], ds[16];
    const char* sK = img;
    const char* sdK = img + 128 * JK;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half == 0) request(1, 1);  // the other 128 keys' k / dk ...
        else request(2, 0);            // ... then the first 128 keys' v / dv, in flight under the score products
#pragma unroll
        for (int bl = 0; bl < 8; ++bl) {
            const int blk = half * 8 + bl;
            s[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
            ds[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int off = (bl * 16 + l16) * JK + (32 * ks + 8 * g) * 2;
                const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sK + off));
                const bf16x8 dkf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sdK + off));
                s[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, __builtin_bit_cast(bf16x8, qf[ks]), s[blk], 0, 0, 0);
                ds[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, __builtin_bit_cast(bf16x8, dqf[ks]), ds[blk], 0, 0, 0);
                ds[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dkf, __builtin_bit_cast(bf16x8, qf[ks]), ds[blk], 0, 0, 0);
            }
            // (no fragment reads of a later key block above this point: left free, hipcc lifts them across the unrolled blocks until the
            // 256 registers are gone and spills row addresses instead)
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // every wave is done with this half's images
        deposit(half == 0 ? JK : JV);
        __syncthreads();
    }

    float mx = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < 16; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[blk][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f, rs = 0.f;
    uint4 pfs[8], wfs[8];  // e and W = e o dS as the bf16 B operands of the value products
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        float e[8], wg[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            e[r] = __expf(s[2 * p + (r >> 2)][r & 3] - mx);
            wg[r] = e[r] * ds[2 * p + (r >> 2)][r & 3];
            l += e[r];
            rs += wg[r];
        }
        pfs[p] = make_uint4(pack_bf16(e[0], e[1]), pack_bf16(e[2], e[3]), pack_bf16(e[4], e[5]), pack_bf16(e[6], e[7]));
        wfs[p] = make_uint4(pack_bf16(wg[0], wg[1]), pack_bf16(wg[2], wg[3]), pack_bf16(wg[4], wg[5]), pack_bf16(wg[6], wg[7]));
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);

    f32x4 o[6], u[6], tt[6];
#pragma unroll
    for (int db = 0; db < 6; ++db) {
        o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
        u[db] = f32x4{0.f, 0.f, 0.f, 0.f};
        tt[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char* sV = img;
    const char* sdV = img + 128 * JV;
    // transposed-read address of this lane inside a 4-row x 16-column block: row l16 >> 2, columns 4 (l16 & 3) ..
    const int troff = (4 * g + (l16 >> 2)) * JV + 4 * (l16 & 3) * 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half == 0) request(2, 1);
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            const int p = half * 4 + pl;
            const int rb = 32 * pl * JV + troff;
#pragma unroll
            for (int db = 0; db < 6; ++db) {
                typedef __attribute__((address_space(3))) s16x4* lds4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(sV + rb + db * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(sV + rb + 16 * JV + db * 32));
                const s16x4 dlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(sdV + rb + db * 32));
                const s16x4 dhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(sdV + rb + 16 * JV + db * 32));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                const bf16x8 dvf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(dlo, dhi, 0, 1, 2, 3, 4, 5, 6, 7));
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, pfs[p]), o[db], 0, 0, 0);
                u[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, wfs[p]), u[db], 0, 0, 0);
                tt[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dvf, __builtin_bit_cast(bf16x8, pfs[p]), tt[db], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (half == 0) {
            __syncthreads();
            deposit(JV);
            __syncthreads();
        }
    }

    const float rl = 1.0f / l, rr = rs * rl;
    const int64_t orow = (tok0 + jvp_window_token(a, w, qh * 128 + wv * 16 + l16)) * a.ldo + head * HD;
    bf16_t* po = static_cast<bf16_t*>(a.out) + orow;
    bf16_t* pdo = static_cast<bf16_t*>(a.dout) + orow;
#pragma unroll
    for (int db = 0; db < 6; ++db) {
        const int d = 16 * db + 4 * g;
        if (d < HD) {
            *reinterpret_cast<uint2*>(po + d) = make_uint2(pack_bf16(o[db][0] * rl, o[db][1] * rl), pack_bf16(o[db][2] * rl, o[db][3] * rl));
            float dv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[r] = (u[db][r] + tt[db][r] - rr * o[db][r]) * rl;
            *reinterpret_cast<uint2*>(pdo + d) = make_uint2(pack_bf16(dv[0], dv[1]), pack_bf16(dv[2], dv[3]));
        }
    }
}

// --------------------------------------------------------------------------------- sCM target (loss.py:236-247)
// g = -cos^2 t (sd F - dxt) - r (cos t sin t x_t + sd dF);  g /= (rms_sample(g) + 0.1);  target = F + g
// (the loss kernel then sees (F - target)^2 = g^2 with gradient -2 w g through F only, as Fx - Fx.detach() - g does)
__global__ __launch_bounds__(256) void scm_g_kernel(const float* __restrict__ F, const float* __restrict__ dxt,
                                                    const float* __restrict__ xt_over_sd, const float* __restrict__ dF,
                                                    const float* __restrict__ t, float r, float sd, float* __restrict__ gbuf,
                                                    float* __restrict__ ss, int64_t per) {
    const int b = blockIdx.y;
    const float c = cosf(t[b]), s = sinf(t[b]);
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
        const int64_t k = (int64_t)b * per + i;
        const float gv = -(c * c) * (sd * F[k] - dxt[k]) - r * ((c * s) * (xt_over_sd[k] * sd) + sd * dF[k]);
        gbuf[k] = gv;
        acc += gv * gv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(ss + b, acc);
}
__global__ __launch_bounds__(256) void scm_target_kernel(const float* __restrict__ F, float* __restrict__ gbuf,
                                                         const float* __restrict__ ss, int64_t per) {
    const int b = blockIdx.y;
    const float inv = 1.0f / (sqrtf(ss[b] / (float)per) + 0.1f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
        const int64_t k = (int64_t)b * per + i;
        gbuf[k] = F[k] + gbuf[k] * inv;
    }
}

}  // namespace

int g_modnorm_jvp_rows = 1;  // tuning key 17: swiftk_modnorm_jvp_pair walks 32 n rows per block (0 = a row per wave)

#define DT_SWITCH(dtype, CALL_BF16, CALL_F32) \
    if (dtype == SWIFTK_BF16) { CALL_BF16; } else if (dtype == SWIFTK_F32) { CALL_F32; } else return SWIFTK_EINVAL

extern "C" int swiftk_timestep_embed_jvp(const float* t, const float* dt, const float* freqs, float* demb, int B, int d,
                                         float timestep_weight, void* stream) {
    if (!t || !dt || !freqs || !demb || B <= 0 || d <= 0) return SWIFTK_EINVAL;
    hipLaunchKernelGGL(temb_jvp_kernel, dim3((B * d + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), t, dt, freqs,
                       demb, B, d, timestep_weight);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_silu_jvp(const float* z, const float* dz, float* y, float* dy, int64_t n, void* stream) {
    if (!z || !dz || !dy || n <= 0) return SWIFTK_EINVAL;
    hipLaunchKernelGGL(silu_jvp_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), z, dz, y, dy, n);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

template <int HD>
static int launch_qknorm_jvp(void* qkv, void* dqkv, int64_t ld, const float* scale, float* rn, int64_t M, int heads, int dtype,
                             hipStream_t st) {
    const int grid = grid_for(M * heads * 2 * 16);
    DT_SWITCH(dtype,
              hipLaunchKernelGGL((qknorm_jvp_kernel<bf16_t, HD>), dim3(grid), dim3(256), 0, st, static_cast<bf16_t*>(qkv),
                                 static_cast<bf16_t*>(dqkv), ld, scale, rn, M, heads),
              hipLaunchKernelGGL((qknorm_jvp_kernel<float, HD>), dim3(grid), dim3(256), 0, st, static_cast<float*>(qkv),
                                 static_cast<float*>(dqkv), ld, scale, rn, M, heads));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_qknorm_jvp(void* qkv, void* dqkv, int64_t ld, const float* scale, float* rn, int64_t M, int heads,
                                 int head_dim, int dtype, void* stream) {
    if (!qkv || !dqkv || !scale || M <= 0 || heads <= 0) return SWIFTK_EINVAL;
    if (ld < 3 * heads * head_dim) return SWIFTK_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (head_dim) {
        case 80: return launch_qknorm_jvp<80>(qkv, dqkv, ld, scale, rn, M, heads, dtype, st);
        case 88: return launch_qknorm_jvp<88>(qkv, dqkv, ld, scale, rn, M, heads, dtype, st);
        case 96: return launch_qknorm_jvp<96>(qkv, dqkv, ld, scale, rn, M, heads, dtype, st);
    }
    return SWIFTK_ESHAPE;
}

extern "C" int swiftk_swiglu_jvp(const void* h, const void* dh, int64_t ldh, void* out, void* dout, int64_t ldo, int64_t M,
                                 int mlp, int dtype, void* stream) {
    if (!h || !dh || !out || !dout || M <= 0 || mlp <= 0) return SWIFTK_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(M * mlp);
    DT_SWITCH(dtype,
              hipLaunchKernelGGL(swiglu_jvp_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(h),
                                 static_cast<const bf16_t*>(dh), ldh, static_cast<bf16_t*>(out), static_cast<bf16_t*>(dout),
                                 ldo, M, mlp),
              hipLaunchKernelGGL(swiglu_jvp_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(h),
                                 static_cast<const float*>(dh), ldh, static_cast<float*>(out), static_cast<float*>(dout), ldo,
                                 M, mlp));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_modnorm_jvp(const void* y, const void* dy, int64_t ldy, float* x, float* dx, void* xT, void* dxT,
                                  int64_t ldxT, const float* gamma, const float* beta, const float* mod, const float* dmod,
                                  int64_t ldmod, int64_t M, int d, int64_t rows_per_sample, float eps, int dtype,
                                  void* stream) {
    if (!y || !dy || !x || !dx || !xT || !dxT || !gamma || !beta || !mod || !dmod || M <= 0 || rows_per_sample <= 0)
        return SWIFTK_EINVAL;
    if (d > 1536 || M % rows_per_sample || ldxT < d || ldy < d) return SWIFTK_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((M + 3) / 4);
    const bool vec = !(d & 7) && !(ldy & 7) && !(ldxT & 7) && !(ldmod & 3) && !(((uintptr_t)y | (uintptr_t)dy | (uintptr_t)xT | (uintptr_t)dxT) & 15) &&
                     !(((uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mod | (uintptr_t)dmod) & 15);
#define SWIFTK_MNJ(KERN)                                                                                                       \
    DT_SWITCH(dtype,                                                                                                           \
              hipLaunchKernelGGL(KERN<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(y),                    \
                                 static_cast<const bf16_t*>(dy), ldy, x, dx, static_cast<bf16_t*>(xT), static_cast<bf16_t*>(dxT), \
                                 ldxT, gamma, beta, mod, dmod, ldmod, M, d, rows_per_sample, eps),                             \
              hipLaunchKernelGGL(KERN<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(y),                      \
                                 static_cast<const float*>(dy), ldy, x, dx, static_cast<float*>(xT), static_cast<float*>(dxT), \
                                 ldxT, gamma, beta, mod, dmod, ldmod, M, d, rows_per_sample, eps))
    if (vec) { SWIFTK_MNJ(modnorm_jvp_kernel); } else { SWIFTK_MNJ(modnorm_jvp_scalar_kernel); }
#undef SWIFTK_MNJ
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_modnorm_jvp_pair(const void* y, const void* dy, int64_t ldy, const void* xT_in, const void* dxT_in, void* xT,
                                       void* dxT, int64_t ldxT, void* x_lo, void* dx_lo, const float* gamma, const float* beta,
                                       const float* mod, const float* dmod, int64_t ldmod, int64_t M, int d,
                                       int64_t rows_per_sample, float eps, void* stream) {
    if (!y || !dy || !xT_in || !dxT_in || !xT || !dxT || !x_lo || !dx_lo || !gamma || !beta || !mod || !dmod || M <= 0 ||
        rows_per_sample <= 0)
        return SWIFTK_EINVAL;
    if (d > 1536 || (d & 7) || M % rows_per_sample || ldxT < d || ldy < d || (ldy & 7) || (ldxT & 7) || (ldmod & 3)) return SWIFTK_ESHAPE;
    if (((uintptr_t)y | (uintptr_t)dy | (uintptr_t)xT | (uintptr_t)dxT | (uintptr_t)xT_in | (uintptr_t)dxT_in | (uintptr_t)gamma |
         (uintptr_t)beta | (uintptr_t)mod | (uintptr_t)dmod) & 15)
        return SWIFTK_EALIGN;
    if (((uintptr_t)x_lo | (uintptr_t)dx_lo) & 7) return SWIFTK_EALIGN;
    if (g_modnorm_jvp_rows > 0 && rows_per_sample % (32 * g_modnorm_jvp_rows) == 0) {
        const int rpb = 32 * g_modnorm_jvp_rows;
        hipLaunchKernelGGL(modnorm_jvp_pair_rows_kernel, dim3((unsigned)(M / rpb)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const bf16_t*>(y), static_cast<const bf16_t*>(dy), ldy, static_cast<const bf16_t*>(xT_in),
                           static_cast<const bf16_t*>(dxT_in), static_cast<bf16_t*>(xT), static_cast<bf16_t*>(dxT), ldxT,
                           static_cast<uint8_t*>(x_lo), static_cast<uint8_t*>(dx_lo), gamma, beta, mod, dmod, ldmod, M, d,
                           rows_per_sample, eps, rpb);
        SWIFTK_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(modnorm_jvp_pair_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(y), static_cast<const bf16_t*>(dy), ldy, static_cast<const bf16_t*>(xT_in),
                       static_cast<const bf16_t*>(dxT_in), static_cast<bf16_t*>(xT), static_cast<bf16_t*>(dxT), ldxT,
                       static_cast<uint8_t*>(x_lo), static_cast<uint8_t*>(dx_lo), gamma, beta, mod, dmod, ldmod, M, d, rows_per_sample,
                       eps);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

template <int HD>
static int launch_attn_jvp(const AttnJvpArgs& a, int grid, int dtype, hipStream_t st) {
    DT_SWITCH(dtype, hipLaunchKernelGGL(attn_jvp_bf16_kernel<HD>, dim3(grid), dim3(512), 0, st, a),
              hipLaunchKernelGGL((attn_jvp_kernel<float, HD>), dim3(grid), dim3(512), 0, st, a));
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_window_attention_jvp(const void* qkv, const void* dqkv, int64_t ldq, void* out, void* dout, int64_t ldo,
                                           int B, int gh, int gw, int heads, int head_dim, int shift_h, int shift_w, int dtype,
                                           void* stream) {
    if (!qkv || !dqkv || !out || !dout || B <= 0 || heads <= 0) return SWIFTK_EINVAL;
    if (head_dim != 80 && head_dim != 88 && head_dim != 96) return SWIFTK_ESHAPE;
    if (gh <= 0 || gw <= 0 || gh % 16 || gw % 16) return SWIFTK_ESHAPE;
    if (shift_h < 0 || shift_w < 0 || shift_h >= gh || shift_w >= gw) return SWIFTK_ESHAPE;
    if (ldq < 3 * heads * head_dim || ldo < heads * head_dim) return SWIFTK_ESHAPE;
    const int es = dtype == SWIFTK_BF16 ? 2 : 4;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)dqkv & 15) || (ldq * es) % 16) return SWIFTK_EALIGN;
    AttnJvpArgs a{qkv, dqkv, out, dout, ldq, ldo, gh, gw, heads, shift_h, shift_w, gw / 16, (gh / 16) * (gw / 16)};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = B * a.nw * heads * 2;
    if (dtype == SWIFTK_BF16 && ((uintptr_t)out & 7 || (uintptr_t)dout & 7 || (ldo & 3))) return SWIFTK_EALIGN;
    if (head_dim == 80) return launch_attn_jvp<80>(a, grid, dtype, st);
    if (head_dim == 96) return launch_attn_jvp<96>(a, grid, dtype, st);
    return launch_attn_jvp<88>(a, grid, dtype, st);
}

extern "C" int swiftk_scm_target(const float* F, const float* dxt, const float* xt_over_sd, const float* dF, const float* t,
                                 float r, float sigma_data, float* target, float* ss_scratch, int B, int64_t per_sample,
                                 void* stream) {
    if (!F || !dxt || !xt_over_sd || !dF || !t || !target || !ss_scratch || B <= 0 || per_sample <= 0) return SWIFTK_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (swiftk_zero_f32_impl(ss_scratch, B, stream, 1) != 0) return SWIFTK_EINVAL;  // (a kernel, not hipMemsetAsync: common.h)
    const dim3 grid((unsigned)grid_for(per_sample, 256, 512), (unsigned)B);
    hipLaunchKernelGGL(scm_g_kernel, grid, dim3(256), 0, st, F, dxt, xt_over_sd, dF, t, r, sigma_data, target, ss_scratch,
                       per_sample);
    hipLaunchKernelGGL(scm_target_kernel, grid, dim3(256), 0, st, F, target, ss_scratch, per_sample);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}
