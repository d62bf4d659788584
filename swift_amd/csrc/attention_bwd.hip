// Backward of the shifted-window cosine attention core for gfx950 (bf16, head_dim 80 / 88 / 96, pre-normalised q/k).
//
// Given qh (scaled, normalised q), kh, v, the forward output O and dO, one workgroup per (sample, window, head)
// recomputes P = softmax(qh kh^T) (the whole 256 x 256 problem is on chip) and produces
//     dV = P^T dO,   dP = dO V^T,   dS = P o (dP - delta),  delta_q = dO_q . O_q,   dqh = dS kh,   dkh = dS^T qh.
// Two passes so that no accumulation ever crosses waves:
//   pass A  waves own 32 QUERY rows: S^T[key][q] orientation as in the forward kernel -> row statistics (m, 1/l), delta
//           and dqh^T[d][q] = kh^T dS^T (dS^T accumulators reused as the MFMA B operand);
//   pass B  waves own 32 KEYS: S[q][key] orientation (lane = key), P and dS rebuilt from the saved row statistics ->
//           dv^T[d][key] = dO^T P and dkh^T[d][key] = qh^T dS, again accumulator-as-operand.
// The recomputation costs two extra 256x256x88 products per item (attention is 4 % of the model's FLOPs) and buys
// register-resident dq / dk / dv with plain stores -- no atomics, no cross-wave reduction.
// LDS: two 256 x 208-B images (K,V in pass A; Q,dO in pass B: row reads for the A operands, ds_read_b64_tr_b16 for
// the transposed ones) + 3 KiB of row statistics.
#include "common.h"

namespace {

constexpr int NT = 512;
constexpr int DB = 3;         // 32-row blocks of a transposed head vector (head_dim <= 96)
constexpr int STR = 208;      // image row stride (96 bf16 + 16 B)
constexpr int IMG = 256 * STR;
constexpr float LOG2E = 1.4426950408889634f;

struct BwdArgs {
    const bf16_t* qkvh;   // [B, ntok, ldq]  q-hat | k-hat | v per head
    const bf16_t* o;      // [B, ntok, ldo]  forward output
    const bf16_t* d_o;    // [B, ntok, ldo]  upstream gradient
    bf16_t* dqkvh;        // [B, ntok, ldd]  gradient w.r.t. q-hat | k-hat | v
    int64_t ldq, ldo, ldd;
    int gh, gw, heads, sh, sw, nwx, nw;
};

__device__ __forceinline__ int wtoken(const BwdArgs& a, int w, int j) {
    const int wy = w / a.nwx, wx = w - wy * a.nwx;
    int gy = wy * 16 + (j >> 4) + a.sh;
    int gx = wx * 16 + (j & 15) + a.sw;
    gy = gy >= a.gh ? gy - a.gh : gy;
    gx = gx >= a.gw ? gx - a.gw : gx;
    return gy * a.gw + gx;
}

// copy one head_dim-element row (HD/8 x 16 B) into an image row and zero the tail chunks up to d = 95
template <int HD>
__device__ __forceinline__ void stage_row(char* img, int j, const bf16_t* src) {
    constexpr int NCH = HD / 8;
    uint4 r[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) r[c] = *reinterpret_cast<const uint4*>(src + 8 * c);
#pragma unroll
    for (int c = 0; c < NCH; ++c) *reinterpret_cast<uint4*>(img + j * STR + 16 * c) = r[c];
#pragma unroll
    for (int c = NCH; c < 12; ++c) *reinterpret_cast<uint4*>(img + j * STR + 16 * c) = make_uint4(0, 0, 0, 0);
}

// fragments of a row as the MFMA "B" operand with the row index on the lane: chunk 2*ks + hh (a chunk past the row = zeros)
template <int HD, int KS>
__device__ __forceinline__ void row_frags(const bf16_t* row, int hh, uint4 (&f)[KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (2 * ks + 1 >= HD / 8) {
            const uint4 t = *reinterpret_cast<const uint4*>(row + 8 * (2 * ks));
            f[ks] = hh ? make_uint4(0, 0, 0, 0) : t;
        } else {
            f[ks] = *reinterpret_cast<const uint4*>(row + 8 * (2 * ks + hh));
        }
    }
}

__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
    const uint32_t ua[4] = {a.x, a.y, a.z, a.w}, ub[4] = {b.x, b.y, b.z, b.w};
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s += __uint_as_float(ua[e] << 16) * __uint_as_float(ub[e] << 16);
        s += __uint_as_float(ua[e] & 0xffff0000u) * __uint_as_float(ub[e] & 0xffff0000u);
    }
    return s;
}

__device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A operand = transposed image block: rows (k-slots) base..base+15 of the image, columns d = 32*db + (lane&31)
__device__ __forceinline__ uint4 tr_frag(const char* img, int base_row, int db, int vbase) {
    const char* p = img + base_row * STR + vbase + db * 64;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 8 * STR));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(uint4, v);
}

__device__ __forceinline__ uint4 pack8(const f32x16& s, int s2) {
    uint4 p;
    p.x = pack_bf16(s[8 * s2 + 0], s[8 * s2 + 1]);
    p.y = pack_bf16(s[8 * s2 + 2], s[8 * s2 + 3]);
    p.z = pack_bf16(s[8 * s2 + 4], s[8 * s2 + 5]);
    p.w = pack_bf16(s[8 * s2 + 6], s[8 * s2 + 7]);
    return p;
}

// store a transposed accumulator set acc[db][reg] = X^T[d][row], row = lane&31 -> dst_row[d] (bf16, 8-B pieces)
template <int HD>
__device__ __forceinline__ void store_t(bf16_t* dst_row, const f32x16 (&acc)[DB], int hh) {
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + g * 8 + hh * 4;
            if (d < HD)
                *reinterpret_cast<uint2*>(dst_row + d) = make_uint2(pack_bf16(acc[db][4 * g], acc[db][4 * g + 1]),
                                                                    pack_bf16(acc[db][4 * g + 2], acc[db][4 * g + 3]));
        }
}

template <int HD>
__global__ __launch_bounds__(NT) void attn_bwd_kernel(BwdArgs a) {
    constexpr int KS = (HD + 15) / 16;  // 16-wide k-steps over head_dim: 5 / 6 / 6
    __shared__ __attribute__((aligned(16))) char imgA[IMG];
    __shared__ __attribute__((aligned(16))) char imgB[IMG];
    __shared__ __attribute__((aligned(16))) float st_m[256];
    __shared__ __attribute__((aligned(16))) float st_il[256];
    __shared__ __attribute__((aligned(16))) float st_dl[256];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x % a.heads;
    const int w = (blockIdx.x / a.heads) % a.nw;
    const int b = blockIdx.x / (a.heads * a.nw);
    const int64_t tok0 = (int64_t)b * a.gh * a.gw;
    const int c32 = lane & 31, hh = lane >> 5, i16 = lane & 15;
    const int vbase = (4 * hh + (i16 >> 2)) * STR + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;
    const int my_tok = wtoken(a, w, wv * 32 + c32);  // the query (pass A) / key (pass B) this lane's column stands for
    // this lane's rows are addressed from the token index each time they are needed (through an opaque copy: hipcc would otherwise
    // build the four 64-bit row pointers up here and carry them -- spilled -- through both passes; all the kernel keeps is my_tok)
    auto tok = [&]() {
        int t = my_tok;
        asm volatile("" : "+v"(t));
        return tok0 + t;
    };
#define MY_QKV (a.qkvh + tok() * a.ldq + head * 3 * HD)
#define MY_O (a.o + tok() * a.ldo + head * HD)
#define MY_DO (a.d_o + tok() * a.ldo + head * HD)
#define MY_DQKV (a.dqkvh + tok() * a.ldd + head * 3 * HD)

    // ---------------------------------------------------------------- pass A images: K -> imgA, V -> imgB
    if (tid < 256) {
        const bf16_t* src = a.qkvh + (tok0 + wtoken(a, w, tid)) * a.ldq + head * 3 * HD;
        stage_row<HD>(imgA, tid, src + HD);
        stage_row<HD>(imgB, tid, src + 2 * HD);
    }
    uint4 qf[KS], dof[KS];
    row_frags<HD, KS>(MY_QKV, hh, qf);
    row_frags<HD, KS>(MY_DO, hh, dof);
    float delta;
    {
        uint4 of[KS];
        row_frags<HD, KS>(MY_O, hh, of);
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s += dot8(dof[ks], of[ks]);
        delta = s + __shfl_xor(s, 32, 64);
    }
    __syncthreads();

    f32x16 s[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            s[kb] = mfma(*reinterpret_cast<const uint4*>(imgA + (kb * 32 + c32) * STR + ks * 32 + hh * 16), qf[ks], s[kb]);
    }
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
            const float p = __builtin_amdgcn_exp2f(s[kb][r] * LOG2E - mb);
            s[kb][r] = p;
            l += p;
        }
    l += __shfl_xor(l, 32, 64);
    const float il = 1.0f / l;
    if (hh == 0) {
        st_m[wv * 32 + c32] = mx;
        st_il[wv * 32 + c32] = il;
        st_dl[wv * 32 + c32] = delta;
    }
    // dS^T = P^T o (dP^T - delta),  dP^T[key][q] = V[key][:] . dO[q][:];  then dqh^T[d][q] += kh^T[d][key] dS^T[key][q]
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            dp = mfma(*reinterpret_cast<const uint4*>(imgB + (kb * 32 + c32) * STR + ks * 32 + hh * 16), dof[ks], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = s[kb][r] * il * (dp[r] - delta);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const uint4 ds = pack8(s[kb], s2);
#pragma unroll
            for (int db = 0; db < DB; ++db) dq[db] = mfma(tr_frag(imgA, kb * 32 + s2 * 16, db, vbase), ds, dq[db]);
        }
    }
    store_t<HD>(MY_DQKV, dq, hh);
    __syncthreads();

    // ---------------------------------------------------------------- pass B images: Q -> imgA, dO -> imgB
    if (tid < 256) {
        int etid = tid;  // (opaque: the staging token is recomputed here, not carried from pass A's staging)
        asm volatile("" : "+v"(etid));
        const int t = wtoken(a, w, etid);
        stage_row<HD>(imgA, etid, a.qkvh + (tok0 + t) * a.ldq + head * 3 * HD);
        stage_row<HD>(imgB, etid, a.d_o + (tok0 + t) * a.ldo + head * HD);
    }
    uint4 kf[KS], vf[KS];
    {
        const bf16_t* r = MY_QKV;
        row_frags<HD, KS>(r + HD, hh, kf);
        row_frags<HD, KS>(r + 2 * HD, hh, vf);
    }
    __syncthreads();

    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[db][r] = dv[db][r] = 0.f;
#pragma unroll 1
    for (int qb = 0; qb < 8; ++qb) {
        // S[q][key] = Q K^T (rows q in registers, this lane's key on the column), dP[q][key] = dO V^T
        f32x16 sq, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sq[r] = dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            sq = mfma(*reinterpret_cast<const uint4*>(imgA + (qb * 32 + c32) * STR + ks * 32 + hh * 16), kf[ks], sq);
            dp = mfma(*reinterpret_cast<const uint4*>(imgB + (qb * 32 + c32) * STR + ks * 32 + hh * 16), vf[ks], dp);
        }
        f32x16 pp, ds;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q0 = qb * 32 + 8 * g + 4 * hh;  // rows of registers 4g..4g+3
            const float4 m4 = *reinterpret_cast<const float4*>(st_m + q0);
            const float4 i4 = *reinterpret_cast<const float4*>(st_il + q0);
            const float4 d4 = *reinterpret_cast<const float4*>(st_dl + q0);
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __builtin_amdgcn_exp2f((sq[4 * g + e] - mm[e]) * LOG2E) * ii[e];
                pp[4 * g + e] = p;
                ds[4 * g + e] = p * (dp[4 * g + e] - dd[e]);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const uint4 pf = pack8(pp, s2), df = pack8(ds, s2);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                dv[db] = mfma(tr_frag(imgB, qb * 32 + s2 * 16, db, vbase), pf, dv[db]);  // dO^T[d][q] P[q][key]
                dk[db] = mfma(tr_frag(imgA, qb * 32 + s2 * 16, db, vbase), df, dk[db]);  // qh^T[d][q] dS[q][key]
            }
        }
    }
    {
        bf16_t* r = MY_DQKV;
        store_t<HD>(r + HD, dk, hh);
        store_t<HD>(r + 2 * HD, dv, hh);
    }
#undef MY_QKV
#undef MY_O
#undef MY_DO
#undef MY_DQKV
}


// ------------------------------------------------------------------------------------------------------------------
// Persistent form for head_dim 88 (round 3).  The kernel above spends 45 us per item for 10.5 us of MFMA time: one
// workgroup per item, images staged through registers by 256 of its 512 threads, every load latency exposed four times
// per item (K/V images, fragment rows, Q/dO images, fragment rows) with one workgroup per CU.  Here a fixed grid walks the
// item list (XCD-concurrent order of attention_pipe.hip) and every image travels by LDS-DMA into one of THREE 45-KB
// buffers (row-major 176-B rows, lane-linear pieces), rotated so that two of an item's four image loads hide under
// compute: Q(n) lands under pass A, K(n+1) under pass B.  Pass A no longer keeps S^T of the whole window in registers
// (128 VGPRs): a first sweep over the key blocks finds the row maximum (skipped when the logit bound exp(min(scale, ln 100))
// <= 48 lets offset 0 stand in), a second sweep rebuilds S^T block by block and accumulates
//     l += e,   dq^T += kh^T [e o (dP^T - delta)],     e = exp(S^T - m),
// with the 1/l factor -- a per-lane scalar, a lane owns one query column -- applied to dq once at the end.
// Round 6: templated on head_dim -- 80 (the 468 M variant: 160-B rows, five whole k-steps, 40 pieces per image, 146,560 B of LDS) and
// 96 (the 664 M variant: 192-B rows, 48 pieces; three 48-KB images leave room for 8-row output slabs only: four staging rounds per
// accumulator set instead of two, 162,944 B) beside 88.  Their row strides are not conflict-free as 176 B happens to be (40 / 48
// dwords: 2- / 4-way on the 16-B row reads); the images still travel by lane-linear DMA, which rules a padded stride out.
template <int HD>
struct PipeGeom {
    static constexpr int ROW = 2 * HD, TILE = 256 * ROW;  // image: 256 window rows of one head vector, row-major
    static constexpr int KS = (HD + 15) / 16;             // 16-wide k-steps over head_dim: 5 / 5.5 / 6
    static constexpr int CPR = HD / 8;                    // 16-B chunks per row
    static constexpr int NPQ = TILE / 1024;               // 1-KiB DMA pieces per image: 40 / 44 / 48
    static constexpr int NPW = (NPQ + 7) / 8;             // piece slots per wave
    static constexpr int OR = HD > 88 ? 8 : 16;           // rows per round of the wave-private output staging
    static constexpr int OSLAB = OR * ROW;
    static constexpr int NSTORE = (32 / OR) * ((OR * CPR + 63) / 64);  // dwordx4 stores a wave issues per accumulator set
    static constexpr int LDS = 3 * TILE + 64 + 3 * 1024 + 8 * OSLAB + 64;  // 88: 160,896 B (the last 64: per-head logit-scale gradient sums)
};

template <int ROW>
__device__ __forceinline__ uint4 tr_frag_row(const char* img, int base_row, int db, int vbase) {
    const char* p = img + base_row * ROW + vbase + db * 64;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 8 * ROW));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(uint4, v);
}

struct BwdPArgs {
    BwdArgs a;
    const float* scale;  // per-head logit scale parameter (bounds |logit|); null = unknown (always take the maximum)
    int nitems;
    // QK-norm backward folded into the stores (swiftk_window_attention_bwd_qknorm): rn [tokens, 3 heads] = 1 / max(|.|, 1e-12) of the raw
    // q / k vectors, dscale [heads] accumulates d(logit scale); null = the gradients of q-hat / k-hat leave as they are
    const float* rn;
    float* dscale;
    int dbg;  // timing experiments (tuning key 4, bits 16..): 1 no pass-A sweeps, 2 no pass-B loop, 4 no output stores, 8 no max sweep, 16 Q image requested at the item's top
};

// Backward of the cosine-attention prologue (SWIFTK_EPI_QKNORM: x-hat = tau x / n) on a transposed accumulator set, in place:
// acc[db][4 g + e] = d(x-hat)[d][row] with d = 32 db + 8 g + 4 hh + e and the row on the lane.  The row's x-hat values are at
// hand as the MFMA B-operand fragments f[ks] = 16-B chunk 2 ks + hh of the row; the accumulator layout wants elements
// 4 hh .. 4 hh + 3 of chunk 4 db + g, i.e. for even g the low (hh = 0) or high (hh = 1) half of the chunk the hh = 0 lane holds
// and for odd g the same of the hh = 1 lane's chunk: one exchange of two dwords per k-step with lane ^ 32.
//   dx = rn (tau d(x-hat) - x-hat (x-hat . d(x-hat)) / tau);   returns x-hat . d(x-hat) (= tau x d(tau)'s share of this row)
template <int HD, int KS>
__device__ __forceinline__ float qknorm_bwd_acc(f32x16 (&acc)[DB], const uint4 (&f)[KS], int hh, float tau, float rn) {
    uint32_t up[DB][4][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int db = ks >> 1, ge = 2 * (ks & 1);
        const uint32_t r0 = __shfl_xor(hh ? f[ks].x : f[ks].z, 32, 64), r1 = __shfl_xor(hh ? f[ks].y : f[ks].w, 32, 64);
        up[db][ge][0] = hh ? r0 : f[ks].x;
        up[db][ge][1] = hh ? r1 : f[ks].y;
        up[db][ge + 1][0] = hh ? f[ks].z : r0;
        up[db][ge + 1][1] = hh ? f[ks].w : r1;
    }
    float dot = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (db * 32 + g * 8 >= HD) continue;  // (rows d >= 88 of the accumulators are never stored; their contents are arbitrary)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w = up[db][g][e >> 1];
                dot = fmaf(acc[db][4 * g + e], __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16)), dot);
            }
        }
    dot += __shfl_xor(dot, 32, 64);
    const float ca = rn * tau, cb = -rn * dot / tau;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (db * 32 + g * 8 >= HD) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w = up[db][g][e >> 1];
                acc[db][4 * g + e] = fmaf(cb, __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16)), ca * acc[db][4 * g + e]);
            }
        }
    return dot;
}

template <int HD>
__global__ __launch_bounds__(NT) void attn_bwd_pipe_kernel(BwdPArgs pa) {
    using G = PipeGeom<HD>;
    constexpr int KS = G::KS, CPR = G::CPR, NPQ = G::NPQ, ROWB = G::ROW, TILEB = G::TILE, OSLABB = G::OSLAB, OR = G::OR, BWD_LDS = G::LDS;
    const BwdArgs& a = pa.a;
    __shared__ __attribute__((aligned(16))) char smem[BWD_LDS];
    float* st_m = reinterpret_cast<float*>(smem + 3 * TILEB + 64);
    float* st_il = st_m + 256;
    float* st_dl = st_m + 512;
    float* st_ds = reinterpret_cast<float*>(smem + BWD_LDS - 64);  // [16] d(logit scale) sums of this workgroup (zeroed with the rest)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, hh = lane >> 5, i16 = lane & 15;
    const int vbase = (4 * hh + (i16 >> 2)) * ROWB + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;
    char* oslab = smem + 3 * TILEB + 64 + 3 * 1024 + wv * OSLABB;
    const int64_t ntok = (int64_t)a.gh * a.gw;
    // A transposed accumulator set X^T[d][row] (row = this lane's c32 of the wave's 32 window rows) leaves through the
    // wave's LDS slab, 16 rows per round: written as the 8-B pieces the MFMA layout yields, read back as 16-B chunks of
    // consecutive row bytes, so one dwordx4 store covers ~5.8 whole 176-B row segments.  The row-per-lane form (22 8-byte
    // stores per lane, 64 different rows per instruction) put 4,224 partial-line writes per wave and item on the CU's
    // memory path -- more than the kernel's arithmetic costs.
    auto store_rows = [&](const f32x16 (&acc)[DB], int w_, int64_t tok0_, int col0) {
#pragma unroll
        for (int rnd = 0; rnd < 32 / OR; ++rnd) {
            if ((c32 / OR) == rnd) {
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d = db * 32 + g * 8 + hh * 4;
                        if (d < HD)
                            *reinterpret_cast<uint2*>(oslab + (c32 & (OR - 1)) * ROWB + d * 2) =
                                make_uint2(pack_bf16(acc[db][4 * g], acc[db][4 * g + 1]), pack_bf16(acc[db][4 * g + 2], acc[db][4 * g + 3]));
                    }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < (OR * CPR + 63) / 64; ++t) {
                const int c = lane + 64 * t;
                if (c < OR * CPR) {
                    const int row = c / CPR, cc = c - row * CPR;
                    const uint4 v = *reinterpret_cast<const uint4*>(oslab + row * ROWB + cc * 16);
                    const int tok = wtoken(a, w_, wv * 32 + rnd * OR + row);
                    *reinterpret_cast<uint4*>(a.dqkvh + (tok0_ + tok) * a.ldd + col0 + cc * 8) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    const int64_t ldq_b = a.ldq * 2, ldo_b = a.ldo * 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

    int first, last, istep;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, nx = gridDim.x >> 3;
        first = (int)((int64_t)xcd * pa.nitems / 8) + (blockIdx.x >> 3);
        last = (int)((int64_t)(xcd + 1) * pa.nitems / 8);
        istep = nx;
    } else {
        first = (int)((int64_t)blockIdx.x * pa.nitems / gridDim.x);
        last = (int)((int64_t)(blockIdx.x + 1) * pa.nitems / gridDim.x);
        istep = 1;
    }
    if (first >= last) return;
    for (int o = tid * 16; o < BWD_LDS; o += NT * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    auto decode = [&](int item, int& b, int& w, int& h) {
        h = item % a.heads;
        const int r = item / a.heads;
        w = r % a.nw;
        b = r / a.nw;
    };
    auto pin = [&](const char* base) {
        const uint64_t u = (uint64_t)base;
        return (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(u >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)u));
    };
    // An image = 44 pieces of 1 KiB; this wave issues pieces wv + 8 i, i = 0..5 (slots 44..47 repeat pieces 0..3: uniform
    // counts).  Chunk c = 64 p + lane of the image is 16-B chunk c % 11 of window row c / 11; the per-lane source offsets are
    // recomputed for every request (a few dozen VALU instructions per image) rather than held in twelve registers.
    auto dma_img = [&](int buf, const char* base, int w_, int64_t ld_b) {
        const char* bs = pin(base);
#pragma unroll
        for (int i = 0; i < G::NPW; ++i) {
            int p = wv + 8 * i;
            p = p >= NPQ ? p - NPQ : p;
            const int c = p * 64 + lane;
            const int row = c / CPR, cc = c - row * CPR;
            dma_piece(lds0 + buf * TILEB + p * 1024, bs, (uint32_t)(wtoken(a, w_, row) * (int)ld_b) + 16u * cc);
        }
    };
    auto qkv_base = [&](int b, int h, int part) {
        return reinterpret_cast<const char*>(a.qkvh) + (int64_t)b * ntok * ldq_b + (int64_t)(h * 3 + part) * ROWB;
    };
    auto do_base = [&](int b, int h) { return reinterpret_cast<const char*>(a.d_o) + (int64_t)b * ntok * ldo_b + (int64_t)h * ROWB; };

    int b, w, h;
    decode(first, b, w, h);
    int bK = 0, bV = 1, bF = 2;
    dma_img(bK, qkv_base(b, h, 1), w, ldq_b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dma_img(bV, qkv_base(b, h, 2), w, ldq_b);
    __builtin_amdgcn_s_barrier();  // the first K image is complete (later ones are published by the end-of-item barrier)

    for (int item = first; item < last; item += istep) {
        int nb = b, nwn = w, nh = h;
        const bool has_next = item + istep < last;
        if (has_next) decode(item + istep, nb, nwn, nh);
        const float bound = pa.scale ? __expf(fminf(pa.scale[h], 4.605170185988092f)) : INFINITY;
        const bool online = !(bound <= 48.f) && !(pa.dbg & 8);
        const int64_t tok0 = (int64_t)b * ntok;

        // ------------------------------------------------------------------------------ pass A: K in bK, V in bV
        // (every wave is past the previous item: the third buffer is free; K(n) is complete, V(n) still landing)
        // this lane's query row of q-hat and dO as MFMA B-operand fragments, and delta = dO . O of that row.  (Loading them
        // one item ahead, in front of the previous item's dk / dv stores, with a counted wait here was measured slower:
        // 571 against 481 us per launch -- 48 more live registers across the item boundary, spills inside the loops.)
        if (pa.dbg & 16) dma_img(bF, qkv_base(b, h, 0), w, ldq_b);  // (A/B: the round-3 position, in front of the row fragments)
        uint4 qf[KS], dof[KS];
        float delta, rnq = 0.f, rnk = 0.f;
        {
            const int64_t t0 = tok0 + wtoken(a, w, wv * 32 + c32);
            if (pa.rn) {  // (window row wv * 32 + c32 is this lane's QUERY in pass A and its KEY in pass B: one token)
                rnq = pa.rn[t0 * (3 * a.heads) + 3 * h];
                rnk = pa.rn[t0 * (3 * a.heads) + 3 * h + 1];
            }
            row_frags<HD, KS>(a.qkvh + t0 * a.ldq + h * 3 * HD, hh, qf);
            row_frags<HD, KS>(a.d_o + t0 * a.ldo + h * HD, hh, dof);
            uint4 of[KS];
            row_frags<HD, KS>(a.o + t0 * a.ldo + h * HD, hh, of);
            float sd = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) sd += dot8(dof[ks], of[ks]);
            delta = sd + __shfl_xor(sd, 32, 64);
        }
        const char* imK = smem + bK * TILEB;
        const char* imV = smem + bV * TILEB;
        float mx = 0.f;
        if (online) {
            mx = -INFINITY;
#pragma unroll 1
            for (int kb = 0; kb < 8; ++kb) {
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    sc = mfma(*reinterpret_cast<const uint4*>(imK + (kb * 32 + c32) * ROWB + ks * 32 + hh * 16), qf[ks], sc);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of V have landed (under the maximum sweep) ...
        __builtin_amdgcn_s_barrier();                     // ... every wave's have
        // Q image for pass B: requested only now, so that this item's row fragments above did not queue behind its 45 KB
        // (VMEM returns in order); it has the whole of pass A to land.  (An L2 look-ahead for the NEXT item's row fragments -- one
        // 4-byte LDS-DMA per lane and tensor during pass B -- was measured too: 456 -> 466 us, like every such look-ahead here.)
        if (!(pa.dbg & 16)) dma_img(bF, qkv_base(b, h, 0), w, ldq_b);
        const float mb = mx * LOG2E;
        float l = 0.f;
        f32x16 dq[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
#pragma unroll 1
        for (int kb = (pa.dbg & 1) ? 8 : 0; kb < 8; ++kb) {
            f32x16 sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                sc = mfma(*reinterpret_cast<const uint4*>(imK + (kb * 32 + c32) * ROWB + ks * 32 + hh * 16), qf[ks], sc);
                dp = mfma(*reinterpret_cast<const uint4*>(imV + (kb * 32 + c32) * ROWB + ks * 32 + hh * 16), dof[ks], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(sc[r] * LOG2E - mb);
                l += e;
                sc[r] = e * (dp[r] - delta);  // l x dS^T: the 1/l factor is applied to dq at the end
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint4 ds = pack8(sc, s2);
#pragma unroll
                for (int db = 0; db < DB; ++db) dq[db] = mfma(tr_frag_row<ROWB>(imK, kb * 32 + s2 * 16, db, vbase), ds, dq[db]);
            }
        }
        l += __shfl_xor(l, 32, 64);
        const float il = 1.0f / l;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[db][r] *= il;
        if (hh == 0) {
            st_m[wv * 32 + c32] = mx;
            st_il[wv * 32 + c32] = il;
            st_dl[wv * 32 + c32] = delta;
        }

        // ------------------------------------------------------------------------------ pass B: Q in bF, dO -> bK
        // this lane's key row of K and V as MFMA B-operand fragments: straight from the images, which are still intact
        uint4 kf[KS], vf[KS];
        {
            const char* krow = imK + (wv * 32 + c32) * ROWB;
            const char* vrow = imV + (wv * 32 + c32) * ROWB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks == KS - 1 && (HD & 15)) {  // head_dim 88 = 5.5 k-steps: the last one has one real 16-B chunk
                    const uint4 tk = *reinterpret_cast<const uint4*>(krow + (2 * ks) * 16);
                    const uint4 tv = *reinterpret_cast<const uint4*>(vrow + (2 * ks) * 16);
                    kf[ks] = hh ? make_uint4(0, 0, 0, 0) : tk;
                    vf[ks] = hh ? make_uint4(0, 0, 0, 0) : tv;
                } else {
                    kf[ks] = *reinterpret_cast<const uint4*>(krow + (2 * ks + hh) * 16);
                    vf[ks] = *reinterpret_cast<const uint4*>(vrow + (2 * ks + hh) * 16);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // every wave is done with the K and V images; the row statistics are written
        dma_img(bK, do_base(b, h), w, ldo_b);
        if (pa.rn) {
            // d(q-hat) -> dq, and this item's share of d(logit scale) = sum over the window's queries of q-hat . d(q-hat)
            // (every row is counted by both lane halves; no gradient where the clamp at ln 100 is active, swinv2.py:125)
            const float dot = qknorm_bwd_acc<HD, KS>(dq, qf, hh, expf(fminf(pa.scale[h], 4.605170185988092f)), rnq);
            const float tot = 0.5f * wave_sum(dot);
            if (lane == 0 && pa.scale[h] < 4.605170185988092f) atomicAdd(st_ds + h, tot);
        }
        if (!(pa.dbg & 4)) {
            store_rows(dq, w, tok0, h * 3 * HD);  // (the dO image lands under the dq stores)
            // VMEM retires in issue order and the dO pieces are older than the six row stores: leaving exactly those outstanding
            // is enough for the image -- the stores drain under pass B instead of in front of it
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NSTORE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // Q (requested a pass ago) and dO images complete
        if (has_next) {                // the next item's K image lands under pass B
            dma_img(bV, qkv_base(nb, nh, 1), nwn, ldq_b);
        }
        const char* imQ = smem + bF * TILEB;
        const char* imO = smem + bK * TILEB;
        f32x16 dk[DB], dv[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dk[db][r] = dv[db][r] = 0.f;
#pragma unroll 1
        for (int qb = (pa.dbg & 2) ? 8 : 0; qb < 8; ++qb) {
            f32x16 sq, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) sq[r] = dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                sq = mfma(*reinterpret_cast<const uint4*>(imQ + (qb * 32 + c32) * ROWB + ks * 32 + hh * 16), kf[ks], sq);
                dp = mfma(*reinterpret_cast<const uint4*>(imO + (qb * 32 + c32) * ROWB + ks * 32 + hh * 16), vf[ks], dp);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // P and dS overwrite S and dP in place (this phase runs at the register limit)
                const int q0 = qb * 32 + 8 * g + 4 * hh;
                const float4 m4 = *reinterpret_cast<const float4*>(st_m + q0);
                const float4 i4 = *reinterpret_cast<const float4*>(st_il + q0);
                const float4 d4 = *reinterpret_cast<const float4*>(st_dl + q0);
                const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f((sq[4 * g + e] - mm[e]) * LOG2E) * ii[e];
                    sq[4 * g + e] = p;
                    dp[4 * g + e] = p * (dp[4 * g + e] - dd[e]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint4 pf = pack8(sq, s2), df = pack8(dp, s2);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    dv[db] = mfma(tr_frag_row<ROWB>(imO, qb * 32 + s2 * 16, db, vbase), pf, dv[db]);
                    dk[db] = mfma(tr_frag_row<ROWB>(imQ, qb * 32 + s2 * 16, db, vbase), df, dk[db]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next K image (it had the whole pass) ...
        __builtin_amdgcn_s_barrier();  // ... every wave's; and every wave is done with the Q and dO images
        if (has_next) dma_img(bF, qkv_base(nb, nh, 2), nwn, ldq_b);  // next V over the dead Q image: lands under the stores below
        if (pa.rn) qknorm_bwd_acc<HD, KS>(dk, kf, hh, 1.0f, rnk);    // d(k-hat) -> dk
        if (!(pa.dbg & 4)) {                                 // and under the next item's maximum sweep
            store_rows(dk, w, tok0, h * 3 * HD + HD);
            store_rows(dv, w, tok0, h * 3 * HD + 2 * HD);
        }
        // next item: K sits where this item's V was, V where Q was; this item's K / dO buffer is the free one
        const int t = bK;
        bK = bV;
        bV = bF;
        bF = t;
        b = nb; w = nwn; h = nh;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pa.dscale) {
        __syncthreads();
        if (tid < a.heads && st_ds[tid] != 0.f) atomicAdd(pa.dscale + tid, st_ds[tid]);
    }
}

}  // namespace

int g_attn_bwd_fuse = 1;                   // tuning key 15 (A/B): 0 = the QK-norm backward stays a second pass
int g_attn_bwd_pipe = 1;                   // tuning key 9 (A/B): 0 = one workgroup per item (the round-1 kernel)
static bool ntok_bytes_ok(const BwdArgs& a) {    // per-lane 32-bit source offsets inside one sample
    return (int64_t)a.gh * a.gw * a.ldq * 2 < (1ll << 32) && (int64_t)a.gh * a.gw * a.ldo * 2 < (1ll << 32);
}

extern "C" int swiftk_window_attention_bwd(const void* qkvh, int64_t ldq, const void* o, const void* d_o, int64_t ldo,
                                           void* dqkvh, int B, int gh, int gw, int heads, int head_dim, int shift_h,
                                           int shift_w, int dtype, void* stream) {
    return swiftk_window_attention_bwd_scaled(qkvh, ldq, o, d_o, ldo, dqkvh, ldq, nullptr, B, gh, gw, heads, head_dim, shift_h, shift_w,
                                              dtype, stream);
}

// rn != null: the QK-norm backward rides in the stores where the pipelined kernel runs (*fused = 1), else the caller follows up
static int attn_bwd_impl(const void* qkvh, int64_t ldq, const void* o, const void* d_o, int64_t ldo, void* dqkvh, int64_t ldd,
                         const float* scale, const float* rn, float* dscale, int* fused, int B, int gh, int gw, int heads,
                         int head_dim, int shift_h, int shift_w, int dtype, void* stream) {
    if (fused) *fused = 0;
    if (!qkvh || !o || !d_o || !dqkvh || B <= 0 || heads <= 0) return SWIFTK_EINVAL;
    if (dtype != SWIFTK_BF16) return SWIFTK_ESHAPE;  // training runs under bf16 autocast (trainer.py:191)
    if (head_dim != 80 && head_dim != 88 && head_dim != 96) return SWIFTK_ESHAPE;
    if (gh <= 0 || gw <= 0 || gh % 16 || gw % 16) return SWIFTK_ESHAPE;
    if (shift_h < 0 || shift_w < 0 || shift_h >= gh || shift_w >= gw) return SWIFTK_ESHAPE;
    if (ldq < 3 * heads * head_dim || ldo < heads * head_dim || ldd < 3 * heads * head_dim) return SWIFTK_ESHAPE;
    if (((uintptr_t)qkvh & 15) || ((uintptr_t)o & 15) || ((uintptr_t)d_o & 15) || ((uintptr_t)dqkvh & 15) || (ldq * 2) % 16 ||
        (ldo * 2) % 16 || (ldd * 2) % 16)
        return SWIFTK_EALIGN;
    BwdArgs a;
    a.qkvh = static_cast<const bf16_t*>(qkvh);
    a.o = static_cast<const bf16_t*>(o);
    a.d_o = static_cast<const bf16_t*>(d_o);
    a.dqkvh = static_cast<bf16_t*>(dqkvh);
    a.ldq = ldq;
    a.ldo = ldo;
    a.ldd = ldd;
    a.gh = gh;
    a.gw = gw;
    a.heads = heads;
    a.sh = shift_h;
    a.sw = shift_w;
    a.nwx = gw / 16;
    a.nw = (gh / 16) * (gw / 16);
    const dim3 grid(B * a.nw * heads);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_attn_bwd_pipe && ntok_bytes_ok(a)) {
        BwdPArgs pa;
        pa.a = a;
        pa.scale = scale;
        pa.nitems = B * a.nw * heads;
        pa.rn = nullptr;
        pa.dscale = nullptr;
        if (rn && dscale && scale && heads <= 16 && g_attn_bwd_fuse) {
            pa.rn = rn;
            pa.dscale = dscale;
            *fused = 1;
        }
        pa.dbg = g_attn_dbg >> 16;
        int pgrid = 256;
        if (pa.nitems < pgrid) pgrid = pa.nitems >= 8 ? (pa.nitems & ~7) : pa.nitems;
        if (head_dim == 80) hipLaunchKernelGGL(attn_bwd_pipe_kernel<80>, dim3(pgrid), dim3(NT), 0, st, pa);
        else if (head_dim == 96) hipLaunchKernelGGL(attn_bwd_pipe_kernel<96>, dim3(pgrid), dim3(NT), 0, st, pa);
        else hipLaunchKernelGGL(attn_bwd_pipe_kernel<88>, dim3(pgrid), dim3(NT), 0, st, pa);
        SWIFTK_CHECK_LAUNCH();
        return 0;
    }
    if (head_dim == 80) hipLaunchKernelGGL(attn_bwd_kernel<80>, grid, dim3(NT), 0, st, a);
    else if (head_dim == 96) hipLaunchKernelGGL(attn_bwd_kernel<96>, grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(attn_bwd_kernel<88>, grid, dim3(NT), 0, st, a);
    SWIFTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int swiftk_window_attention_bwd_scaled(const void* qkvh, int64_t ldq, const void* o, const void* d_o, int64_t ldo,
                                                  void* dqkvh, int64_t ldd, const float* scale, int B, int gh, int gw, int heads,
                                                  int head_dim, int shift_h, int shift_w, int dtype, void* stream) {
    return attn_bwd_impl(qkvh, ldq, o, d_o, ldo, dqkvh, ldd, scale, nullptr, nullptr, nullptr, B, gh, gw, heads, head_dim, shift_h,
                         shift_w, dtype, stream);
}

extern "C" int swiftk_window_attention_bwd_qknorm(const void* qkvh, int64_t ldq, const void* o, const void* d_o, int64_t ldo,
                                                  void* dqkv, int64_t ldd, const float* scale, const float* rn, float* dscale, int B,
                                                  int gh, int gw, int heads, int head_dim, int shift_h, int shift_w, int dtype,
                                                  void* stream) {
    if (!scale || !rn || !dscale) return SWIFTK_EINVAL;
    int fused = 0;
    const int rc = attn_bwd_impl(qkvh, ldq, o, d_o, ldo, dqkv, ldd, scale, rn, dscale, &fused, B, gh, gw, heads, head_dim, shift_h,
                                 shift_w, dtype, stream);
    if (rc != 0 || fused) return rc;
    // head_dim 80 / 96, or the one-workgroup-per-item kernel: the gradients of q-hat / k-hat are rewritten in place by a second pass
    return swiftk_qknorm_bwd(qkvh, dqkv, ldq, rn, dqkv, ldd, scale, dscale, (int64_t)B * gh * gw, heads, head_dim, dtype, stream);
}
