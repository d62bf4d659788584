// Shifted-window cosine attention for gfx950, one workgroup per (sample, window, head).
//
// Fused in one launch: cyclic-roll + window gather (pure address arithmetic, the
// rolled/partitioned tensor of swinv2.py:193-198 is never materialised), L2
// normalisation of q and k, the per-head logit scale, S = q k^T over the
// window's 256 keys, a single-pass row softmax (the whole row is on chip, so no
// online rescaling), O = P V and the scatter back to token order.
//
// MFMA orientation: S^T[key][q] = K Q^T so that a lane owns ONE query column
// (its 256 logits sit in 128 registers of lanes l and l^32): the row max / sum
// are register reductions plus one cross-half exchange.  The S^T accumulator
// is then, register for register, the B operand of O^T[d][q] = V^T P^T
// (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's
// operand"); V^T fragments come from a row-major V image through
// ds_read_b64_tr_b16 with the k order permuted to match.
//
// bf16: 8 waves x 32 query rows, v_mfma_f32_32x32x16_bf16, K/Q/V images in LDS (152 KiB).
// fp32: 8 waves x 32 query rows, v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), K then V share one LDS image.
#include "common.h"

namespace {

constexpr int WTOK = 256;  // 16 x 16 window
constexpr int NT = 512;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN100 = 4.605170185988092f;  // ln(1/0.01), swinv2.py:125

struct AttnArgs {
    const void* qkv;
    void* out;
    const float* scale;
    int64_t ldq, ldo;
    int gh, gw, heads, sh, sw, nwx, nw;
    int prenorm;  // q, k arrive L2-normalised (and q scaled) from the to_qkv GEMM epilogue
    int pv3;      // fp32 kernel: O = P V as three bf16 products of split operands (SWIFTK_ATTN_PV_BF16X3)
};

// token index (row-major, un-rolled grid) of window-local token j of window w: roll(-s)[p] = x[(p+s) mod n]
__device__ __forceinline__ int window_token(const AttnArgs& a, int w, int j) {
    const int wy = w / a.nwx, wx = w - wy * a.nwx;
    int gy = wy * 16 + (j >> 4) + a.sh;
    int gx = wx * 16 + (j & 15) + a.sw;
    gy = gy >= a.gh ? gy - a.gh : gy;
    gx = gx >= a.gw ? gx - a.gw : gx;
    return gy * a.gw + gx;
}

// ------------------------------------------------------------------------------------------------ bf16

constexpr int KSTR = 208;  // bytes per K/Q row: 96 bf16 + 16 pad -> conflict-free ds_read_b128 over 16 rows
constexpr int VSTR = 192;  // bytes per V row: 96 bf16 -> conflict-free ds_read_b64_tr_b16 over 4 rows x 64 B

template <int HD>
__global__ __launch_bounds__(NT) void attn_bf16_kernel(AttnArgs a) {
    static_assert(HD % 8 == 0 && HD <= 96, "head_dim");
    constexpr int NCH = HD / 8;          // 16-B chunks per head vector
    constexpr int KS = (HD + 15) / 16;   // k-steps of the 32x32x16 MFMA over d
    constexpr int DB = (HD + 31) / 32;   // 32-wide output blocks over d
    __shared__ __attribute__((aligned(16))) char sK[WTOK * KSTR];
    __shared__ __attribute__((aligned(16))) char sQ[WTOK * KSTR];
    __shared__ __attribute__((aligned(16))) char sV[WTOK * VSTR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x % a.heads;
    const int w = (blockIdx.x / a.heads) % a.nw;
    const int b = blockIdx.x / (a.heads * a.nw);
    const int64_t tok0 = (int64_t)b * a.gh * a.gw;

    // ---- phase A: gather + normalise into LDS.  threads 0..255: k and v of token tid; 256..511: q of token tid-256
    {
        const int j = tid & 255;
        const bf16_t* src = static_cast<const bf16_t*>(a.qkv) + (tok0 + window_token(a, w, j)) * a.ldq + head * 3 * HD;
        if (tid >= 256) {
            uint4 raw[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) raw[c] = *reinterpret_cast<const uint4*>(src + c * 8);
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t* u = reinterpret_cast<const uint32_t*>(&raw[c]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(u[e] << 16), hi = __uint_as_float(u[e] & 0xffff0000u);
                    ss += lo * lo + hi * hi;
                }
            }
            const float tau = expf(fminf(a.scale[head], LN100));
            const float inv = a.prenorm ? 1.0f : tau / fmaxf(sqrtf(ss), 1e-12f);
            char* dst = sQ + j * KSTR;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t* u = reinterpret_cast<const uint32_t*>(&raw[c]);
                uint4 o;
                uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ou[e] = pack_bf16(__uint_as_float(u[e] << 16) * inv, __uint_as_float(u[e] & 0xffff0000u) * inv);
                *reinterpret_cast<uint4*>(dst + c * 16) = o;
            }
#pragma unroll
            for (int c = NCH; c < 12; ++c) *reinterpret_cast<uint4*>(dst + c * 16) = make_uint4(0, 0, 0, 0);
        } else {
            uint4 kr[NCH], vr[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) kr[c] = *reinterpret_cast<const uint4*>(src + HD + c * 8);
#pragma unroll
            for (int c = 0; c < NCH; ++c) vr[c] = *reinterpret_cast<const uint4*>(src + 2 * HD + c * 8);
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t* u = reinterpret_cast<const uint32_t*>(&kr[c]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(u[e] << 16), hi = __uint_as_float(u[e] & 0xffff0000u);
                    ss += lo * lo + hi * hi;
                }
            }
            const float inv = a.prenorm ? 1.0f : 1.0f / fmaxf(sqrtf(ss), 1e-12f);
            char* dk = sK + j * KSTR;
            char* dv = sV + j * VSTR;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t* u = reinterpret_cast<const uint32_t*>(&kr[c]);
                uint4 o;
                uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ou[e] = pack_bf16(__uint_as_float(u[e] << 16) * inv, __uint_as_float(u[e] & 0xffff0000u) * inv);
                *reinterpret_cast<uint4*>(dk + c * 16) = o;
                *reinterpret_cast<uint4*>(dv + c * 16) = vr[c];
            }
#pragma unroll
            for (int c = NCH; c < 12; ++c) {
                *reinterpret_cast<uint4*>(dk + c * 16) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(dv + c * 16) = make_uint4(0, 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // ---- phase B: this wave owns query rows 32*wv .. 32*wv+31
    const int c32 = lane & 31, hh = lane >> 5;
    uint4 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const uint4*>(sQ + (wv * 32 + c32) * KSTR + ks * 32 + hh * 16);

    f32x16 s[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4 kf = *reinterpret_cast<const uint4*>(sK + (kb * 32 + c32) * KSTR + ks * 32 + hh * 16);
            // D[key][q] += K[key][d] * Q[q][d]
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[ks]),
                                                            s[kb], 0, 0, 0);
        }
    }

    // softmax over the 256 keys of query column c32: 128 values here, 128 in lane^32
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
    const float mb = mx * LOG2E;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = exp2f(s[kb][r] * LOG2E - mb);
            s[kb][r] = p;
            l += p;
        }
    l += __shfl_xor(l, 32, 64);

    // O^T[d][q] += V^T[d][key] * P^T[key][q]; k-slot (hh, j) of step (kb, s2) <-> key 32kb + 16s2 + 8(j>>2) + 4hh + (j&3)
    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    // transposed-read address: 16-lane group g = lane>>4 covers d0 = 16*(g&1); lane 4q+p of the group addresses
    // key row q, columns 4p..4p+3 and receives column (lane&15) of the four rows
    const int i16 = lane & 15;
    const int vbase = (4 * hh + (i16 >> 2)) * VSTR + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint4 pf;
            pf.x = pack_bf16(s[kb][8 * s2 + 0], s[kb][8 * s2 + 1]);
            pf.y = pack_bf16(s[kb][8 * s2 + 2], s[kb][8 * s2 + 3]);
            pf.z = pack_bf16(s[kb][8 * s2 + 4], s[kb][8 * s2 + 5]);
            pf.w = pack_bf16(s[kb][8 * s2 + 6], s[kb][8 * s2 + 7]);
            const char* vrow = sV + (kb * 32 + s2 * 16) * VSTR + vbase;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(vrow + db * 64));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(vrow + db * 64 + 8 * VSTR));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf),
                                                                o[db], 0, 0, 0);
            }
        }
    }

    // ---- scatter: lane holds O[q = c32][d = 32db + 8g + 4hh + 0..3] in registers 4g..4g+3 of o[db]
    const float rl = 1.0f / l;
    bf16_t* dst = static_cast<bf16_t*>(a.out) + (tok0 + window_token(a, w, wv * 32 + c32)) * a.ldo + head * HD;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + g * 8 + hh * 4;
            if (d < HD)
                *reinterpret_cast<uint2*>(dst + d) = make_uint2(pack_bf16(o[db][4 * g] * rl, o[db][4 * g + 1] * rl),
                                                                pack_bf16(o[db][4 * g + 2] * rl, o[db][4 * g + 3] * rl));
        }
}

// ------------------------------------------------------------------------------------------------ fp32

constexpr int VSTR32 = 96;  // floats per V row

// PV3 (round 6, the split engine): S = q k^T and the softmax stay on the exact-fp32 MFMA -- the logits are where operand rounding is
// amplified -- but O = P V runs as THREE bf16 MFMA products of (hi, lo)-split operands, P_hi V_hi + P_lo V_hi + P_hi V_lo (the dropped
// P_lo V_lo term is 2^-18 relative), at 16 / 3 times the fp32 matrix rate: the V image is staged as two bf16 images (hi, lo; the
// bf16 kernel's 192-B rows and transposed reads), the probabilities are split where they are packed into operands.
template <int HD, bool PV3>
__global__ __launch_bounds__(NT) void attn_f32_kernel(AttnArgs a) {
    static_assert(HD % 8 == 0 && HD <= 96, "head_dim (fp32 path)");
    constexpr int KSTR32 = HD + 4;      // floats per K row: 16 consecutive rows land on 16 distinct 4-bank groups
    constexpr int ISTR32 = KSTR32 > VSTR32 ? KSTR32 : VSTR32;
    constexpr int HALF = HD / 2;        // k-slot (kk, hh) <-> d = kk + HALF*hh
    constexpr int NQ = HALF / 4;        // float4 per half row
    constexpr int NF4 = HD / 4;
    constexpr int DB = (HD + 31) / 32;
    __shared__ __attribute__((aligned(16))) float sKV[WTOK * ISTR32];  // K image (stride HD + 4), later V image (stride 96)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x % a.heads;
    const int w = (blockIdx.x / a.heads) % a.nw;
    const int b = blockIdx.x / (a.heads * a.nw);
    const int64_t tok0 = (int64_t)b * a.gh * a.gw;
    const float* qkv = static_cast<const float*>(a.qkv);
    const int c32 = lane & 31, hh = lane >> 5;

    // K image: threads 0..255 normalise one key row each
    if (tid < 256) {
        const float* src = qkv + (tok0 + window_token(a, w, tid)) * a.ldq + head * 3 * HD + HD;
        float4 kr[NF4];
#pragma unroll
        for (int c = 0; c < NF4; ++c) kr[c] = *reinterpret_cast<const float4*>(src + 4 * c);
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NF4; ++c) ss += kr[c].x * kr[c].x + kr[c].y * kr[c].y + kr[c].z * kr[c].z + kr[c].w * kr[c].w;
        const float nrm = a.prenorm ? 1.0f : fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int c = 0; c < NF4; ++c)
            *reinterpret_cast<float4*>(sKV + tid * KSTR32 + 4 * c) =
                make_float4(kr[c].x / nrm, kr[c].y / nrm, kr[c].z / nrm, kr[c].w / nrm);
    }
    // Q: lane (c32, hh) keeps d in [HALF*hh, HALF*hh + HALF) of query row 32*wv + c32, normalised and scaled
    float qreg[HALF];
    {
        const int tq = window_token(a, w, wv * 32 + c32);
        const float* src = qkv + (tok0 + tq) * a.ldq + head * 3 * HD + HALF * hh;
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * c);
            qreg[4 * c] = v.x; qreg[4 * c + 1] = v.y; qreg[4 * c + 2] = v.z; qreg[4 * c + 3] = v.w;
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        ss += __shfl_xor(ss, 32, 64);
        const float nrm = a.prenorm ? 1.0f : fmaxf(sqrtf(ss), 1e-12f);
        const float tau = a.prenorm ? 1.0f : expf(fminf(a.scale[head], LN100));
#pragma unroll
        for (int k = 0; k < HALF; ++k) qreg[k] = qreg[k] / nrm * tau;
    }
    __syncthreads();

    f32x16 s[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        const float* krow = sKV + (kb * 32 + c32) * KSTR32 + HALF * hh;
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(krow + 4 * c);
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[4 * c + 0], s[kb], 0, 0, 0);
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[4 * c + 1], s[kb], 0, 0, 0);
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[4 * c + 2], s[kb], 0, 0, 0);
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[4 * c + 3], s[kb], 0, 0, 0);
        }
    }

    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(s[kb][r] - mx);
            s[kb][r] = p;
            l += p;
        }
    l += __shfl_xor(l, 32, 64);

    // swap the LDS image K -> V
    __syncthreads();
    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    if constexpr (PV3) {
        static_assert(2 * WTOK * VSTR <= (int)sizeof(sKV), "the two bf16 V images must fit into the fp32 K / V buffer");
        char* imH = reinterpret_cast<char*>(sKV);
        char* imL = imH + WTOK * VSTR;
        if (tid < 256) {
            const float* src = qkv + (tok0 + window_token(a, w, tid)) * a.ldq + head * 3 * HD + 2 * HD;
#pragma unroll
            for (int c = 0; c < VSTR / 8; ++c) {  // 8-B pieces of the 192-B rows: HD / 4 of data, the rest zero
                uint2 h = make_uint2(0u, 0u), lo = make_uint2(0u, 0u);
                if (c < NF4) {
                    const float4 v = *reinterpret_cast<const float4*>(src + 4 * c);
                    h = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
                    lo = make_uint2(pack_bf16(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u)),
                                    pack_bf16(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u)));
                }
                *reinterpret_cast<uint2*>(imH + tid * VSTR + 8 * c) = h;
                *reinterpret_cast<uint2*>(imL + tid * VSTR + 8 * c) = lo;
            }
        }
        __syncthreads();
        // O^T[d][q] += V^T[d][key] P^T[key][q]: the S^T accumulators are the B operand, V^T fragments by transposed reads with the k
        // order permuted to match (attn_bf16_kernel's scheme), once per image
        const int i16 = lane & 15;
        const int vbase = (4 * hh + (i16 >> 2)) * VSTR + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;
        typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint32_t ph[4], pl[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p0 = s[kb][8 * s2 + 2 * e], p1 = s[kb][8 * s2 + 2 * e + 1];
                    ph[e] = pack_bf16(p0, p1);
                    pl[e] = pack_bf16(p0 - __uint_as_float(ph[e] << 16), p1 - __uint_as_float(ph[e] & 0xffff0000u));
                }
                const uint4 pH = make_uint4(ph[0], ph[1], ph[2], ph[3]), pL = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                const int roff = (kb * 32 + s2 * 16) * VSTR + vbase;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    auto frag = [&](const char* im) {
                        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(im + roff + db * 64));
                        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(im + roff + db * 64 + 8 * VSTR));
                        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
                    };
                    const bf16x8 vH = frag(imH), vL = frag(imL);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vH, __builtin_bit_cast(bf16x8, pH), o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vH, __builtin_bit_cast(bf16x8, pL), o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vL, __builtin_bit_cast(bf16x8, pH), o[db], 0, 0, 0);
                }
            }
        }
    } else {
    if (tid < 256) {
        const float* src = qkv + (tok0 + window_token(a, w, tid)) * a.ldq + head * 3 * HD + 2 * HD;
#pragma unroll
        for (int c = 0; c < NF4; ++c)
            *reinterpret_cast<float4*>(sKV + tid * VSTR32 + 4 * c) = *reinterpret_cast<const float4*>(src + 4 * c);
#pragma unroll
        for (int c = NF4; c < VSTR32 / 4; ++c)
            *reinterpret_cast<float4*>(sKV + tid * VSTR32 + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2);  // + 4*hh per lane half
            const float* vrow = sKV + (key + 4 * hh) * VSTR32 + c32;
#pragma unroll
            for (int db = 0; db < DB; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[db * 32], s[kb][r], o[db], 0, 0, 0);
        }
    }

    const float rl = 1.0f / l;
    float* dst = static_cast<float*>(a.out) + (tok0 + window_token(a, w, wv * 32 + c32)) * a.ldo + head * HD;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + g * 8 + hh * 4;
            if (d < HD)
                *reinterpret_cast<float4*>(dst + d) =
                    make_float4(o[db][4 * g] * rl, o[db][4 * g + 1] * rl, o[db][4 * g + 2] * rl, o[db][4 * g + 3] * rl);
        }
}

template <int HD>
int launch_hd(const AttnArgs& a, int B, int dtype, hipStream_t st) {
    const int grid = B * a.nw * a.heads;
    if (dtype == SWIFTK_BF16)
        hipLaunchKernelGGL(attn_bf16_kernel<HD>, dim3(grid), dim3(NT), 0, st, a);
    else if (a.pv3)
        hipLaunchKernelGGL((attn_f32_kernel<HD, true>), dim3(grid), dim3(NT), 0, st, a);
    else
        hipLaunchKernelGGL((attn_f32_kernel<HD, false>), dim3(grid), dim3(NT), 0, st, a);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int swiftk_window_attention(const void* qkv, int64_t ldq, void* out, int64_t ldo, const float* scale, int B,
                                       int gh, int gw, int heads, int head_dim, int shift_h, int shift_w, int dtype,
                                       int flags, void* stream) {
    const bool prenorm = flags & SWIFTK_ATTN_PRENORM;
    if (!qkv || !out || (!scale && !prenorm) || B <= 0 || heads <= 0) return SWIFTK_EINVAL;
    if (dtype != SWIFTK_F32 && dtype != SWIFTK_BF16) return SWIFTK_EINVAL;
    if (gh <= 0 || gw <= 0 || gh % 16 || gw % 16) return SWIFTK_ESHAPE;
    if (shift_h < 0 || shift_w < 0 || shift_h >= gh || shift_w >= gw) return SWIFTK_ESHAPE;
    if (ldq < 3 * heads * head_dim || ldo < heads * head_dim) return SWIFTK_ESHAPE;
    const int es = dtype == SWIFTK_BF16 ? 2 : 4;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || (ldq * es) % 16 || (ldo * es) % 16) return SWIFTK_EALIGN;
    AttnArgs a;
    a.qkv = qkv;
    a.out = out;
    a.scale = scale;
    a.ldq = ldq;
    a.ldo = ldo;
    a.gh = gh;
    a.gw = gw;
    a.heads = heads;
    a.sh = shift_h;
    a.sw = shift_w;
    a.nwx = gw / 16;
    a.nw = (gh / 16) * (gw / 16);
    a.prenorm = prenorm ? 1 : 0;
    a.pv3 = (dtype == SWIFTK_F32 && (flags & SWIFTK_ATTN_PV_BF16X3)) ? 1 : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (prenorm && dtype == SWIFTK_BF16 && (head_dim == 80 || head_dim == 88 || head_dim == 96) &&
        !(flags & SWIFTK_ATTN_NO_PIPE)) {
        AttnPipeArgs pa{qkv, out, ldq, ldo, B, gh, gw, heads, shift_h, shift_w, head_dim == 88 ? g_attn_dbg : 0, scale,
                        (flags & SWIFTK_ATTN_TILED) ? 1 : 0, head_dim};
        const bool timed = swiftk_prof_begin(SWIFTK_PROF_ATTENTION, 0, st);
        const int rc = swiftk_launch_attn_pipe(pa, st);
        if (timed) swiftk_prof_end(st);
        return rc;
    }
    if (flags & SWIFTK_ATTN_TILED) return SWIFTK_ESHAPE;  // window-tiled input exists for the pipelined kernel only
    const bool timed = swiftk_prof_begin(SWIFTK_PROF_ATTENTION, 0, st);  // bench.py's attention leg (fp32 engine)
    int rc;
    switch (head_dim) {
        case 96: rc = launch_hd<96>(a, B, dtype, st); break;
        case 88: rc = launch_hd<88>(a, B, dtype, st); break;
        case 80: rc = launch_hd<80>(a, B, dtype, st); break;
        case 64: rc = launch_hd<64>(a, B, dtype, st); break;
        default: rc = SWIFTK_ESHAPE;
    }
    if (timed) swiftk_prof_end(st);
    return rc;
}
