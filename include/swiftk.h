/*
 * swiftk.h -- C ABI of the MI355X (gfx950) kernels behind swift_amd.
 *
 * The reference (stockeh/swift) is pure Python/PyTorch and has no FFI of its
 * own; its "operator boundary" is the Python call
 *     AbstractNetwork.forward(x, t, auxiliary, jvp, return_logvar)
 *                                   (reference: src/swift/models/abstract.py:26-35,
 *                                    src/swift/models/swinv2.py:305-330)
 * reached through PassPrecond.forward (src/swift/models/precond.py:133-148).
 * This header is what a maintainer of the reference would bind with ctypes to
 * replace the ATen ops on that path (INTEGRATION.md shows the stub).  Each
 * entry point below cites the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     the name ends in _host;
 *   - the caller owns all buffers (PyTorch allocations, tensor.data_ptr());
 *     nothing here allocates, synchronises or throws;
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, a negative SWIFTK_E* code on a rejected argument,
 *     or a positive hipError_t if the launch itself failed;
 *   - dtype: SWIFTK_F32 = 0 (IEEE fp32 operands, fp32 MFMA), SWIFTK_BF16 = 1
 *     (bf16 operands, fp32 accumulate).  Normalisation, softmax statistics and
 *     the residual stream are fp32 in both modes.
 */
#ifndef SWIFTK_H
#define SWIFTK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWIFTK_F32 0
#define SWIFTK_BF16 1
#define SWIFTK_BF16X3 2 /* swiftk_model.dtype only: fp32 activations, every GEMM as three bf16 products (swiftk_split3) */

#define SWIFTK_EINVAL (-1)   /* bad argument (null pointer, negative size) */
#define SWIFTK_ESHAPE (-2)   /* shape not supported by the gfx950 kernels   */
#define SWIFTK_EALIGN (-3)   /* pointer / leading dimension mis-aligned     */
#define SWIFTK_EWORKSPACE (-4) /* workspace too small                        */

/* GEMM epilogues */
#define SWIFTK_EPI_NONE 0      /* C = A W^T                                              */
#define SWIFTK_EPI_BIAS_POS 1  /* C = A W^T + bias[n] + pos[(m % pos_rows)][n]           */
#define SWIFTK_EPI_SWIGLU 2    /* C[m][j] = silu(acc[m][2j]) * acc[m][2j+1]  (W rows interleaved gate/up) */
#define SWIFTK_EPI_QKNORM 3    /* to_qkv: per token and head, q <- q/max(|q|,1e-12)*exp(min(ep0[h],ln100)),
                                  k <- k/max(|k|,1e-12), v unchanged (swinv2.py:123-127); ep0 = scale[heads]; head_dim
                                  (80 / 88 / 96; with fp32 operands 80 / 96 need M, N % 8 == 0) travels in `pos_rows`, 0 = 88.
                                  pos_rows = -head_dim (fp32 operands and output, whole tiles): the [q | k]-ONLY form -- W still points at
                                  [q | k | v] row triples and C at [q | k | v] column triples, but N counts 2 head_dim columns per head:
                                  the v rows are skipped and the v columns of C left untouched (the split engine's hot head pairs:
                                  only q-hat and k-hat meet in the logits; one 352-wide tile column per pair instead of two) */
#define SWIFTK_EPI_SWIGLU_BOTH 6 /* training forward of the FeedForward (swinv2.py:96-101): C = A W^T (bf16, the pre-activation the
                                  backward pass needs) AND C2[m][j] = silu(C[m][2j]) * C[m][2j+1]; C2 (bf16) = ep1, its row
                                  stride (elements) = pos_rows; bf16 operands only */
#define SWIFTK_EPI_SWIGLU_BWD 7 /* backward through silu(gate) * up inside the GEMM that produces d(hidden) = dY W2 (autograd of
                                  swinv2.py:100-101): with H = ep1 (bf16 [M, pos_rows], the saved pre-activation, gate/up
                                  interleaved) C[m][2j] = acc * up * (s + gate s (1 - s)), C[m][2j+1] = acc * gate * s,
                                  s = sigmoid(gate); C is bf16 [M, >= 2N]; bf16 operands only */
#define SWIFTK_EPI_ACCUM 5     /* C += A W^T, fp32 C only: the backward pass adds a branch's input gradient onto the
                                  residual-stream gradient (autograd of x + f(x), swinv2.py:211-212) in the GEMM itself */

#define SWIFTK_EPI_SWIGLU_SPLIT3 10 /* the split engine's w1 (bf16 operands = swiftk_split3 blocks, bf16 out): silu(gate) * up leaves as the
                                  NEXT GEMM's operand blocks [hi | lo | hi], hi = bf16(h), lo = bf16(h - hi), block width (columns)
                                  in `pos_rows` (>= N / 2, ldc >= 2 pos_rows + N / 2): what SWIFTK_EPI_SWIGLU with fp32 output +
                                  swiftk_split3(order 0) leave, without h itself reaching memory */
#define SWIFTK_EPI_QKNORM_JVP 8 /* swiftk_gemm_jvp only: SWIFTK_EPI_QKNORM on the primal rows AND its tangent on the tangent rows */
#define SWIFTK_EPI_SWIGLU_JVP 9 /* swiftk_gemm_jvp only: silu(gate) * up and its tangent (pre-activations optionally kept) */

/* swiftk_window_attention flags */
#define SWIFTK_ATTN_PRENORM 1  /* q, k in `qkv` are already normalised / scaled (SWIFTK_EPI_QKNORM); `scale` (optional) bounds
                                  |logit| <= exp(min(scale, ln 100)): where <= 48 softmax needs no row maximum */
#define SWIFTK_ATTN_TILED 4    /* `qkv` is window-tiled (swiftk_gemm_qkv_tiled): [B][window][head][q|k|v][256][head_dim] bf16, windows
                                  of the grid rolled by (shift_h, shift_w); needs PRENORM, bf16, head_dim 88; ldq unused */
#define SWIFTK_ATTN_PV_BF16X3 8 /* fp32 operands (the split engine): the logits and the softmax stay exact fp32, O = P V runs as three bf16 MFMA
                                  products of (hi, lo)-split operands (the dropped lo x lo term: 2^-18 relative) */
#define SWIFTK_ATTN_NO_PIPE 2  /* tuning: keep the one-workgroup-per-item kernel even where the pipelined one applies */

/* Most (member, IC) units one swiftk_swinv2_forward call takes (BASELINE configs[3] puts 96 on a GPU). */
#define SWIFTK_MAX_UNITS 256

int swiftk_version(void);

/* Round `k` up to the K granularity of the GEMM for `dtype` (64 bf16 / 32 fp32 elements = 128 B). */
int64_t swiftk_gemm_k_pad(int dtype, int64_t k);

/*
 * C[M,N] = epilogue(A[M,K] * W[N,K]^T): nn.Linear with its weight as stored.
 * Replaces F.linear at src/swift/models/swinv2.py:119 (to_qkv), :137 (wo),
 * :99-100 (w1 + SwiGLU, w2), :230 (patch embedding) and :240 (head).
 *   A  [M, lda]  dtype, row major, K multiple of swiftk_gemm_k_pad granularity
 *                (pad columns must be finite; W's pad columns must be zero)
 *   W  [N, ldw]  dtype
 *   C  [M, ldc]  out_dtype (SWIFTK_F32 or same as dtype); SWIGLU writes N/2 columns
 *   ep0 = bias[N] fp32, ep1 = pos[pos_rows, N] fp32 (BIAS_POS only)
 */
int swiftk_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                int64_t K, int dtype, int out_dtype, int epilogue, const float* ep0, const float* ep1, int64_t pos_rows,
                void* stream);

/*
 * swiftk_gemm for fp32 operands with TWO-LEVEL accumulation: the MFMA chain restarts every `chunk_k` k (a multiple of 32)
 * and the partial sums meet in fp32 through `scratch` (caller-owned, >= swiftk_gemm_chunk_scratch_bytes(), 16-B aligned;
 * a workgroup-private slab that stays cache-resident).  One chain over K = 1056 .. 2816 ends 1.8 x further from the fp64
 * product than ATen's blocked CPU sgemm (tests/fp32_bisect.py); chains of 256 bring the exact-fp32 engine to the reference's
 * own fp32 accuracy.  Same epilogues, same results up to summation order.  Replaces F.linear of
 * src/swift/models/swinv2.py:119,137,99-100,229,239 in the exact-fp32 engine.
 */
int swiftk_gemm_chunked(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                        int64_t K, int dtype, int out_dtype, int epilogue, const float* ep0, const float* ep1,
                        int64_t pos_rows, int chunk_k, void* scratch, int64_t scratch_bytes, void* stream);
int64_t swiftk_gemm_chunk_scratch_bytes(void);

/*
 * Shifted-window cosine attention, fused: window gather (+cyclic roll), L2
 * normalisation of q and k, per-head logit scale exp(min(s, ln 100)), softmax
 * over the 256 keys of the window, P V, scatter back to token order.
 * Replaces src/swift/models/swinv2.py:189-208 (roll / window_partition /
 * window_reverse / roll) and :120-136 (Attention core).
 *   qkv   [B, gh*gw, ldq] dtype, per token and head h the channels
 *         [h*3*hd, h*3*hd + 3*hd) hold q | k | v (swinv2.py:120-121)
 *   out   [B, gh*gw, ldo] dtype, channels h*hd + d
 *   scale [heads] fp32 (the nn.Parameter, un-exponentiated)
 * Supported: 16x16 windows, head_dim in {64, 80, 88}.  flags: SWIFTK_ATTN_* (bf16 + head_dim 88 + PRENORM
 * runs the persistent LDS-DMA-pipelined kernel).
 */
int swiftk_window_attention(const void* qkv, int64_t ldq, void* out, int64_t ldo, const float* scale, int B, int gh, int gw,
                            int heads, int head_dim, int shift_h, int shift_w, int dtype, int flags, void* stream);

/*
 * x += LayerNorm(y; gamma, beta, eps) * (1 + scale_b) + shift_b, fused with the
 * residual add; also writes a `dtype` copy of the new x as the next GEMM operand.
 * Replaces src/swift/models/swinv2.py:83-86 (ModulatedNorm) + :211-212 (residual).
 *   y     [M, ldy] dtype                 x   [M, d] fp32 (in/out)
 *   xcopy [M, ldc] dtype (may be NULL)   mod [B, ldmod] fp32: scale at [0,d), shift at [d,2d)
 *   rows_per_sample: M / B
 */
int swiftk_modnorm_residual(const void* y, int64_t ldy, float* x, void* xcopy, int64_t ldc, const float* gamma,
                            const float* beta, const float* mod, int64_t ldmod, int64_t M, int d, int64_t rows_per_sample,
                            float eps, int dtype, void* stream);
/* The same update for the split engine (SWIFTK_BF16X3; fp32 y and x): besides x (and the optional fp32 copy) the new rows leave as
 * the NEXT GEMM's operand blocks x3 [M, ld3] bf16 = [hi | lo | hi] over d columns each + zero k-padding -- bit for bit what
 * swiftk_split3(order 0) makes of the fp32 rows, without the pass that re-reads them (10 B per element and a launch less per
 * ModulatedNorm; 4 more where no fp32 copy is needed).  SWIFTK_ESHAPE where the 16-row chunk kernel does not apply
 * (rows_per_sample % 16, d % 8, ld3 - 3 d not a whole number of 16-B chunks up to one k-tile): run the two-step form. */
int swiftk_modnorm_residual_split3(const float* y, int64_t ldy, float* x, float* xcopy, int64_t ldc, void* x3, int64_t ld3,
                                   const float* gamma, const float* beta, const float* mod, int64_t ldmod, int64_t M, int d,
                                   int64_t rows_per_sample, float eps, void* stream);

/*
 * The same update on the bf16 engine's PAIR form of the residual stream: x is held as x_hi = bf16(x), which IS the next
 * GEMM's operand (no separate operand copy is written), plus a low part --
 *   lo_bits 16: x_lo = bf16(x - x_hi)                                  (x to 2^-17 relative; 10 bytes per element and launch)
 *   lo_bits  8: x_lo = one byte, round((x - x_hi) * 256 / ulp(x_hi)) + 128   (x to ulp / 512 = 2^-17 relative;  8 bytes)
 * where the fp32 stream + bf16 copy of swiftk_modnorm_residual move 12 (y 2, x 4 + 4, copy 2).
 * Replaces src/swift/models/swinv2.py:83-86 (ModulatedNorm) + :211-212 (residual) in the bf16 engine.
 *   y [M, ldy] bf16   x_hi [M, ldh] bf16 (in/out; columns >= d untouched)   x_lo [M, ldl] bf16 or uint8 (in/out)
 *   rows_per_sample must be a multiple of 16 (SWIFTK_ESHAPE otherwise: callers keep the fp32-stream form)
 */
int swiftk_modnorm_residual_pair(const void* y, int64_t ldy, void* x_hi, int64_t ldh, void* x_lo, int64_t ldl, int lo_bits,
                                 const float* gamma, const float* beta, const float* mod, int64_t ldmod, int64_t M, int d,
                                 int64_t rows_per_sample, float eps, void* stream);
/* The same with the new hi written to ANOTHER buffer (x_hi_in is left as it was; same row stride): the training forward keeps
 * every layer's operand as a saved activation for the weight gradients.  8-bit low part and d = 1056 / 1280 only
 * (SWIFTK_ESHAPE otherwise: callers keep swiftk_modnorm_residual). */
int swiftk_modnorm_residual_pair_to(const void* y, int64_t ldy, const void* x_hi_in, void* x_hi_out, int64_t ldh, void* x_lo,
                                    int64_t ldl, int lo_bits, const float* gamma, const float* beta, const float* mod,
                                    int64_t ldmod, int64_t M, int d, int64_t rows_per_sample, float eps, void* stream);
/* The same with the branch output given as the SUM of two fp32 slabs y_slabs and y_slabs + slab_stride ([M, ldy] each): what
 * swiftk_gemm_splitk(..., ksplit = 2) leaves.  At one unit per step wo / w2 have 96 output tiles for 256 CUs; two k-ranges
 * per tile fill three quarters of the chip and the norm kernel does the reduction on its way (swiftk_swinv2_forward,
 * tuning key 14). */
int swiftk_modnorm_residual_pair_slabs(const float* y_slabs, int64_t ldy, int64_t slab_stride, void* x_hi, int64_t ldh, void* x_lo,
                                       int64_t ldl, int lo_bits, const float* gamma, const float* beta, const float* mod,
                                       int64_t ldmod, int64_t M, int d, int64_t rows_per_sample, float eps, void* stream);
/* The same with bf16 slabs (swiftk_gemm_splitk_bf16): 10 instead of 14 bytes per element; slab_stride in elements, % 8 == 0. */
int swiftk_modnorm_residual_pair_slabs_bf16(const void* y_slabs, int64_t ldy, int64_t slab_stride, void* x_hi, int64_t ldh, void* x_lo,
                                            int64_t ldl, int lo_bits, const float* gamma, const float* beta, const float* mod,
                                            int64_t ldmod, int64_t M, int d, int64_t rows_per_sample, float eps, void* stream);
/* The pair update of swiftk_modnorm_residual_pair (8-bit low part, d = 1056 or 1280, contiguous y / low-part rows) for a y that
 * arrives as two bf16 k-half slabs [M, d] (slab 1 at y_slabs + slab_stride elements): y = bf16(slab 0 + slab 1), the halves added
 * in fp32 and rounded once, wherever slab 1 exists.  `tail` = NULL: everywhere (behind swiftk_gemm_splitk_bf16); else the three
 * numbers swiftk_gemm_tail_split_bf16 reported (d = 1056 only): slab 1 is read under the tiles that GEMM split and nowhere else.
 * (swinv2.py:77-86 behind :112-113 / :134.) */
int swiftk_modnorm_residual_pair_halves_bf16(const void* y_slabs, int64_t slab_stride, const int64_t* tail, void* x_hi, int64_t ldh,
                                             void* x_lo, const float* gamma, const float* beta, const float* mod, int64_t ldmod,
                                             int64_t M, int d, int64_t rows_per_sample, float eps, void* stream);
/* wo / w2 and the norm above in ONE kernel, for small batches: y = bf16(A[M, K] W[d, K]^T) computed over complete rows (a
 * workgroup owns `rows_per_workgroup` = 32 or 64 rows x all d columns, so the row statistics are there and y never leaves the
 * CU), then the pair update of swiftk_modnorm_residual_pair (8-bit low part).  bf16 operands, d = 1056 or 960, K % 32 == 0,
 * K >= 64 (rows of A and W padded to a multiple of 64 elements; the pad is fetched, never multiplied), M and rows_per_sample
 * multiples of rows_per_workgroup.  Every workgroup streams the whole weight through L2, so
 * this pays while M / rows_per_workgroup is about one round of the CUs (one unit per step at 32 rows, two at 64:
 * swiftk_swinv2_forward, tuning key 23); beyond that swiftk_gemm + swiftk_modnorm_residual_pair is the faster pair.
 * Replaces the same reference lines as those two (swinv2.py:143 / :177 `to_out` / `w2`, :83-86, :211-212). */
int swiftk_gemm_modnorm_residual_pair(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t K, void* x_hi, int64_t ldh,
                                      void* x_lo, int64_t ldl, const float* gamma, const float* beta, const float* mod,
                                      int64_t ldmod, int64_t M, int d, int64_t rows_per_sample, float eps,
                                      int rows_per_workgroup, void* stream);
/* fp32 [rows, lds] -> the pair form: hi [rows, ldh] bf16 with columns [cols, ldh) zeroed (GEMM k-padding), lo [rows, ldl]
 * (bf16 or uint8 by lo_bits, ldl in elements). */
int swiftk_split_pair(const float* src, int64_t lds, void* hi, int64_t ldh, void* lo, int64_t ldl, int lo_bits, int64_t rows,
                      int64_t cols, void* stream);
/* The patch embedding straight into the pair form (round 4): what swiftk_gemm(..., SWIFTK_EPI_BIAS_POS, bias, pos, pos_rows) with fp32
 * output followed by swiftk_split_pair(..., lo_bits 8) leaves -- hi [M, ldh] bf16 (columns [0, N); its k-padding is the caller's),
 * lo [M, ldl] bytes -- without the fp32 stream in between.  bf16 operands, M % 8 == 0, N % 16 == 0, K % 64 == 0. */
int swiftk_gemm_bias_pos_pair(const void* A, int64_t lda, const void* W, int64_t ldw, void* hi, int64_t ldh, void* lo, int64_t ldl,
                              int64_t M, int64_t N, int64_t K, const float* bias, const float* pos, int64_t pos_rows, void* stream);

/*
 * Channel-concat + patchify of up to three NCHW fp32 sources into the GEMM
 * operand of the patch embedding: A[b*gh*gw + gy*gw + gx][(i1*p2 + i2)*C + c]
 * = scale_s * src_s[b][c - c0_s][gy*p1 + i1][gx*p2 + i2]; pad columns zeroed.
 * Replaces src/swift/models/precond.py:139-141 (cat) and swinv2.py:224-229.
 */
int swiftk_patchify(const float* src0, int c0, float s0, const float* src1, int c1, float s1, const float* src2, int c2,
                    float s2, void* A, int64_t lda, int B, int H, int W, int p1, int p2, int dtype, void* stream);

/*
 * out[b][c][y][x] = alpha[b] * xt[b][c][y][x] + beta[b] * tok[b][gy*gw+gx][(c*p1+i1)*p2+i2]
 * (xt may be NULL -> alpha ignored).  Replaces swinv2.py:241-243 (un-patchify)
 * fused with the sampler update cos(t) x_t - sin(t) sigma_d F (diffusion.py:459).
 *   tok [B, gh*gw, ldt] fp32
 */
int swiftk_unpatchify_affine(const float* tok, int64_t ldt, const float* xt, const float* alpha, const float* beta,
                             float* out, int B, int C, int H, int W, int p1, int p2, void* stream);

/*
 * Sinusoidal timestep embedding + auxiliary embedding:
 *   emb[b][i] = sin(t_b w f_i) (i < d/2) | cos(t_b w f_{i-d/2})  + aux_w[i][:] . aux[b][:] sqrt(aux_dim) + aux_b[i]
 * Replaces swinv2.py:44-60 and :318-320.  freqs [d/2] fp32 is supplied by the host.
 */
int swiftk_timestep_embed(const float* t, const float* aux, const float* freqs, const float* aux_w, const float* aux_b,
                          float* emb, int B, int d, int aux_dim, float timestep_weight, void* stream);

/*
 * Small-batch fp32 linear: out[b][n] = act(x[b][:] . W[n][:] + bias[n]) (any B; rows are walked 8 at a time).
 * act: 0 none, 1 SiLU.  Replaces swinv2.py:74 (LatentEmbedding), :85
 * (all modulation Linears, concatenated along n), :327 (logvar).
 */
int swiftk_linear_small(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* out,
                        int64_t ldo, int B, int N, int K, int act, void* stream);

/*
 * Latent noise of one forecast step for B units, as ONE launch: out[b][e] ~ N(0, 1), a pure function of
 * (seeds[b], step, e) -- Philox4x32-10 keyed by the unit's 64-bit seed, counter (e / 4, 0, step lo, step hi), two
 * Box-Muller pairs per counter (24-bit uniforms).  step = (*step_dev if step_dev else 0) + step_add: a captured step
 * graph reads the lead step from device memory and advances it with swiftk_counter_add.
 * Replaces `torch.randn(shape, generator=g)` of src/swift/generating/factory.py:52-56 for production rollouts, where
 * src/swift/generate.py:83 seeds one torch generator per member and consumes it in batch order (so a unit's noise would
 * depend on the sharding); parity tests inject the reference's latents explicitly.
 *   out [B, n_per_unit] fp32 (n_per_unit % 4 == 0)    seeds [B] int64 DEVICE    step_dev DEVICE int64 or NULL
 *   mode 0 = normals; 1 = the generator's raw 32-bit words in the fp32 container (bit-exact test of the integer part)
 */
int swiftk_unit_noise(float* out, const int64_t* seeds, const int64_t* step_dev, int64_t step_add, int B,
                      int64_t n_per_unit, int mode, void* stream);
/* *counter += value on the stream (the lead-step counter of a captured forecast step). */
int swiftk_counter_add(int64_t* counter, int64_t value, void* stream);

/*
 * Residual rollout update in physical units, fused with re-standardisation:
 *   phys[b][c] = xstd[b][c]*sx[c] + mx[c] + y[b][c]*st[c];  xstd[b][c] = (phys[b][c] - mx[c]) / sx[c]
 * Replaces generate.py:120-131 with data/era5.py:110-166.  st == NULL selects the non-residual form of generate.py:132-136
 * (a dataset whose targets are states, not tendencies): phys = y*sx + mx; xstd = y.
 */
int swiftk_rollout_update(float* xstd, const float* y, float* phys, const float* mx, const float* sx, const float* st,
                          int B, int C, int64_t hw, void* stream);

/*
 * Output collection (the reference writes every rank's slice to a shared store,
 * generate.py:139-152, and has no collective): out[b] = fp64 sum of unit b's n fp32 values, added in a
 * fixed order -- bit-identical for a unit whatever rank or batch slot computed it, so ranks can
 * all-gather B doubles per step instead of 9 MB per unit.  scratch: 32*B doubles.
 */
int swiftk_unit_checksum(const float* x, double* out, double* scratch, int B, int64_t n, void* stream);

/*
 * out = a*x + b*y (fp32, out may alias x or y): the samplers' state arithmetic between network
 * evaluations -- re-noising sin(t) sigma_d eps + cos(t) x (diffusion.py:454-455) and the Heun
 * average (diffusion.py:411).
 */
int swiftk_axpby(float* out, float a, const float* x, float b, const float* y, int64_t n, void* stream);

/*
 * p[0..n) = 0.0f by an ORDINARY kernel on `stream`: what every caller uses in front of a kernel that accumulates into p
 * (atomics or read-modify-write: the loss sums of training/loss.py:117-160,306-445, the flat gradient buffer behind
 * trainer.py:219-247's optimizer.zero_grad, the RMSE sums of training/validate.py:96-110).  Never hipMemsetAsync: as a node of a
 * replayed HIP graph it writes a stale fill pattern under the HIP 7.0.x runtime PyTorch bundles (DESIGN section 11;
 * tools/memset_graph_repro.hip); tuning key 25 restores that clear for diagnosis (tools/overflow_campaign.sh).
 */
int swiftk_zero_f32(float* p, int64_t n, void* stream);
/* Diagnosis (tuning key 25 bit 4): out26[0] = checks run, [1..4] = non-zero dwords a clear left behind by dword index mod 4,
 * [5..8] = largest magnitude bits by index mod 4, [9] = samples taken, [10..17] / [18..25] = sample indices / bits. */
int swiftk_zero_check_report(unsigned long long* out26);

/*
 * Measurement hooks (bench.py's roofline leg; not on the reference's path).  After
 * swiftk_profile_gemm(epilogue, N) every swiftk_gemm launch with that epilogue (and that N, if
 * N != 0) is bracketed by a HIP event pair recorded on its launch stream; swiftk_profile_collect
 * synchronises on them, returns the summed kernel time and the launch count, and re-arms.
 * swiftk_profile_gemm(-1, 0) switches the hooks off.  At most 4096 launches per collection.
 * epilogue = SWIFTK_PROF_ATTENTION times the swiftk_window_attention / swiftk_qkv_attention_fused launches instead
 * (N ignored).
 */
#define SWIFTK_PROF_ATTENTION 100
int swiftk_profile_gemm(int epilogue, int64_t N);
/* Tuning knobs (A/B measurements only): key 0 = GEMM variant (0 one tile per workgroup, 1 persistent pipeline),
 * key 1 = tile rows per group of the persistent tile order, key 2 = persistent grid size of the GEMMs and of
 * swiftk_qkv_attention_fused (256 = one workgroup per CU; fewer for a stream created with a CU mask), keys 3 / 4 = ablation
 * bits of the GEMM / attention kernels (timing experiments; results are wrong while set), key 5 = window-tiled q/k/v in
 * swiftk_swinv2_forward (1), key 6 = swiftk_modnorm_residual: bit 0 non-temporal residual-stream accesses, bit 1 chunked
 * kernel (3), key 7 = start-up stagger of the persistent GEMM's workgroups in 1/1000 of an eighth of a tile time (0),
 * key 8 = to_qkv + window attention as one kernel in swiftk_swinv2_forward (1; key 4 bits 8..15 = that kernel's ablations,
 * bits 16.. = those of the persistent attention backward), key 9 = persistent attention backward for head_dim 88 (1),
 * key 12 = residual stream of the bf16 forward: 2 = (bf16 hi, 8-bit lo) pair (default), 1 = (bf16 hi, bf16 lo) pair, 0 = fp32
 * stream + bf16 operand copy,
 * key 13 = chain length (in k) of the fp32-operand GEMMs' two-level accumulation in swiftk_swinv2_forward (256; 0 = off),
 * key 14 = split-K wo / w2 at one unit per step (2 = bf16 slabs, 1 = fp32 slabs, 0 = off; 3 = bf16 slabs beyond one round of the grid as well, an experiment that loses), key 15 = swiftk_window_attention_bwd_qknorm applies the QK-norm backward
 * inside the persistent attention backward (1; 0 = second pass), key 16 = swiftk_modnorm_bwd as one kernel (1; 0 = row pass +
 * column pass; n > 1 = 64 n rows per block), key 17 = swiftk_modnorm_jvp_pair walks 32 n rows per block (1; 0 = a row per wave),
 * key 18 = split engine: w1's epilogue writes w2's (hi, lo) operand blocks itself (1; 0 = fp32 h + swiftk_split3),
 * key 19 = bf16 engine: the patch embedding's epilogue writes the pair form itself (1; 0 = fp32 stream + swiftk_split_pair),
 * key 20 = k-loop of the persistent GEMM with bf16 operands: 1 = ping-pong phases (the SIMD partners alternate between
 * fragment reads + DMA issue and back-to-back MFMAs, counted waits; needs >= 3 k-tiles per work item, else falls back), 0 = one
 * barrier per k-tile (the round-1..4 loop).  Bit-equal results either way.
 * key 21 = the same choice for the k-loop inside swiftk_qkv_attention_fused (1 = ping-pong, 0 = one barrier per k-tile),
 * key 22 = the same for swiftk_gemm_tn_splitk: 1 always (default; measured at all three tile widths), 2 with 352-wide tiles only, 0 never
 *          (384-wide tiles take it at every setting)
 * (Swift-B's four weight gradients 0...-9 % in time, -6 % per layer: tools/tn_ab.py),
 * key 23 = swiftk_swinv2_forward (bf16 engine) runs wo / w2 + norm as swiftk_gemm_modnorm_residual_pair up to this many units
 * per step (0 = never: split-K + slab-summing norm at one unit, GEMM + norm beyond),
 * key 26 = split engine: the fp32 ModulatedNorm writes the next GEMM's operand blocks itself (1; 0 = fp32 copy + swiftk_split3),
 * key 27 = split engine's exact to_qkv recompute: 2 = each hot head alone, q, k and v (default; needs the head mask in
 * swiftk_layer.qk_exact_pairs), 1 = the hot pairs' q and k columns only, 0 = the hot pairs whole,
 * key 28 = split engine: the fp32 attention kernel's P V as three bf16 products (1; 0 = exact fp32 like its q k^T),
 * key 29 = small batches of the bf16 engine (default 3): bit 0 = wo / w2 through swiftk_gemm_tail_split_bf16 where the persistent
 *          walk's last round is at most half full, bit 1 = swiftk_modnorm_residual_pair_halves_bf16 behind the one-unit split-K,
 * key 25 = clears through hipMemsetAsync instead of a kernel (0; diagnosis only; bit 1 = the library's internal clears --
 * swiftk_modnorm_bwd's workspace, swiftk_scm_target's scratch --, bit 2 = swiftk_zero_f32, bit 4 = a check kernel behind
 * swiftk_modnorm_bwd's clear records what it left non-zero: swiftk_zero_check_report). */
int swiftk_set_tuning(int key, int value);
/* The current value of a tuning key (SWIFTK_EINVAL for an unknown key; every valid value is >= 0 or a plain bit mask):
 * what a measurement harness records so that its report names the kernels that actually ran.  Key 11 = the default
 * swiftk_model.x3_exact mask (17). */
int swiftk_get_tuning(int key);
int swiftk_profile_collect(double* total_ms_host, int64_t* launches_host);

/* ------------------------------------------------------------------------ *
 * Training step (reference src/swift/training/trainer.py:189-247, loss.py).
 * The backward pass reuses swiftk_gemm: dgrad = dY * W   as A = dY, "W" = W^T copy;
 * wgrad = dY^T * X as A = dY^T, "W" = X^T, contraction over all tokens, split over
 * workgroups into fp32 slabs (swiftk_gemm_splitk) that swiftk_reduce_slabs sums.
 * ------------------------------------------------------------------------ */

/*
 * to_qkv for the streamed window-attention kernel (swinv2.py:113-121 + window_partition :17-26 + the roll :185-189 in
 * one pass): C = normalise/scale(A W^T) exactly as swiftk_gemm(..., SWIFTK_EPI_QKNORM, scale, ...) computes it, bf16 in
 * and out, N = 3*heads*head_dim, M = B*gh*gw, head_dim 80 / 88 / 96 (the 468 M / Swift-B / 664 M variants of
 * configs/experiment/era5-swinv2-1.4-scm.yaml:21-36), heads even, but stored window-tiled instead of row-major:
 *   qkv_tiled[B][window][head][q|k|v][256][head_dim], window = (ry/16)*(gw/16) + rx/16, index = (ry%16)*16 + rx%16,
 *   (ry, rx) = ((y - shift_h) mod gh, (x - shift_w) mod gw) for the token at grid position (y, x)
 * so that every (window, head) operand of attention is one contiguous, cache-line-aligned 512*head_dim-byte block.
 * Consumed by swiftk_window_attention(..., SWIFTK_ATTN_PRENORM | SWIFTK_ATTN_TILED) with the same shift.
 */
int swiftk_gemm_qkv_tiled(const void* A, int64_t lda, const void* W, int64_t ldw, void* qkv_tiled, int64_t K,
                          const float* scale, int B, int gh, int gw, int heads, int head_dim, int shift_h, int shift_w,
                          void* stream);

/* Batched product: C_b[M, N] = A_b[M, K] * W_b[N, K]^T for b < batch, matrix b of each operand at `stride_*` ELEMENTS from
 * matrix b-1 (bf16 operands, bf16 or fp32 result, K % 64 == 0 with zero-padded rows, N % 4 == 0).  One launch; made for the
 * Newton-Schulz products of MuonWithAuxAdam over the same-shape weights of all layers (reference
 * training/optimizers/muon.py:5-35 runs them as batched torch matmuls too when given a 3-D tensor). */
int swiftk_gemm_batched(const void* A, int64_t lda, int64_t stride_a, const void* W, int64_t ldw, int64_t stride_w, void* C,
                        int64_t ldc, int64_t stride_c, int batch, int64_t M, int64_t N, int64_t K, int dtype, int out_dtype,
                        void* stream);
/* slabs[s][M, ldc] (fp32) = A[M, K_s] * W[N, K_s]^T for the s-th of `ksplit` equal k-ranges; slab s starts at
 * slabs + s*slab_stride.  Same operand rules as swiftk_gemm; M % 8 == 0, N % 8 == 0. */
int swiftk_gemm_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, float* slabs, int64_t ldc, int64_t slab_stride,
                       int64_t M, int64_t N, int64_t K, int dtype, int ksplit, void* stream);
/* Split-K with bf16 slabs (bf16 operands; M, N % 8 == 0): slab s = bf16(A[:, k-range s] W[:, k-range s]^T), each partial product
 * rounded once as a plain bf16 GEMM rounds its output.  The forecast path's wo / w2 at one unit per step. */
int swiftk_gemm_splitk_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* slabs, int64_t ldc, int64_t slab_stride,
                            int64_t M, int64_t N, int64_t K, int ksplit, void* stream);
/* y = A[M, K] W[N, K]^T (bf16, N % 352 == 0: the d-wide Linears wo / w2, swinv2.py:112-113 / :134) for batches whose tile count leaves
 * the persistent walk a last round that is at most half full (T = ceil(M / 256) N / 352 tiles on G = 256 workgroups, 0 < T % G <= G / 2,
 * T > G: 3, 4, 6 .. units per step): the first T - T % G tiles are computed whole into slab 0, the others as two k-halves into slabs 0
 * and 1 (slabs + slab_stride elements), so that every workgroup gets its whole tiles and at most one half -- the last round costs half
 * a tile time.  tail[3] (out) describes where slab 1 exists, for swiftk_modnorm_residual_pair_halves_bf16: the first row of the tile
 * group holding the first split tile (a multiple of 256), the first split tile in the walk's order (groups of tail[2] tile rows, column-
 * major inside a group), and that group height; slab 1 is written under split tiles only.  SWIFTK_ESHAPE when the shape leaves no such
 * round (use swiftk_gemm). */
int swiftk_gemm_tail_split_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* slabs, int64_t ldc, int64_t slab_stride,
                                int64_t M, int64_t N, int64_t K, int64_t* tail, void* stream);
/* The same weight gradient without the transposed copies (TN form; autograd of the Linears at swinv2.py:96-98,112-113,134):
 *   slabs[s][N1, ldc] (fp32) = sum over the s-th of `ksplit` ranges of the K token rows of P[m, 0..N1)^T * Q[m, 0..N2)
 * P = dY [K, ldp], Q = X [K, ldq], bf16, token-major as the forward / backward passes leave them.  K % 64 == 0,
 * N1 % 8 == 0, N2 % 4 == 0; rows must be readable to the end of their last 64-column block: ldp >= roundup(N1, 64),
 * ldq >= roundup(N2, 64) (what lies past N1 / N2 there is never used); output tiles are 352 columns wide where that divides
 * N2, else 320 or 384 (the 352-wide form reads whole tiles: N2 % 352 == 0 or ldq >= roundup(N2, 352)).  SWIFTK_ESHAPE when
 * a shape does not fit (callers then use swiftk_transpose + swiftk_gemm_splitk).  Bit-equal to that path. */
int swiftk_gemm_tn_splitk(const void* P, int64_t ldp, const void* Q, int64_t ldq, float* slabs, int64_t ldc,
                          int64_t slab_stride, int64_t N1, int64_t N2, int64_t K, int ksplit, void* stream);
/* out[r][c] (= | +=) sum_s slabs[s*slab_stride + r*ld_slab + c] */
int swiftk_reduce_slabs(const float* slabs, int64_t ld_slab, int64_t slab_stride, int nslabs, float* out, int64_t ld_out,
                        int64_t rows, int64_t cols, int accumulate, void* stream);
/* dst[c][r] = src[r][c]; dst is [cols, ldd], columns rows..ldd-1 zero-filled */
int swiftk_transpose(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype,
                     void* stream);

/* Un-fused SwiGLU on h [M, 2*mlp] with interleaved (gate_j, up_j) columns, and its backward (swinv2.py:99-100). */
int swiftk_swiglu_fwd(const void* h, int64_t ldh, void* out, int64_t ldo, int64_t M, int mlp, int dtype, void* stream);
int swiftk_swiglu_bwd(const void* h, int64_t ldh, const void* dout, int64_t ldo, void* dh, int64_t lddh, int64_t M, int mlp,
                      int dtype, void* stream);

/* Backward of swiftk_modnorm_residual's norm branch: g = dL/d(out) fp32 [M, d] -> dy (dtype), and fp32 atomic sums
 * dgamma[d], dbeta[d], dmod[B, lddmod] (scale grads at [0,d), shift grads at [d,2d)); the residual branch is identity.
 * row_stats: caller-provided scratch of 2*M floats (per-row mean and 1/std, handed from the row pass to the column pass; the
 * one-kernel form -- rows_per_sample a multiple of 64 and >= d -- keeps its per-sample column sums [2][B][d] there instead). */
int swiftk_modnorm_bwd(const void* y, int64_t ldy, const float* g, void* dy, int64_t lddy, const float* gamma,
                       const float* beta, const float* mod, int64_t ldmod, float* dgamma, float* dbeta, float* dmod,
                       int64_t lddmod, float* row_stats, int64_t M, int d, int64_t rows_per_sample, float eps, int dtype,
                       void* stream);
/* The same for a caller that keeps `workspace` (>= 2 * (M / rows_per_sample) * d floats) ZERO between calls: the one-kernel form
 * only (SWIFTK_ESHAPE where swiftk_modnorm_bwd would take the two-kernel form), without the per-call clear -- the finishing kernel
 * zeroes the sums it has read, so the workspace is zero again on return. */
int swiftk_modnorm_bwd_ws0(const void* y, int64_t ldy, const float* g, void* dy, int64_t lddy, const float* gamma,
                           const float* beta, const float* mod, int64_t ldmod, float* dgamma, float* dbeta, float* dmod,
                           int64_t lddmod, float* workspace, int64_t M, int d, int64_t rows_per_sample, float eps, int dtype,
                           void* stream);

/* Backward of SWIFTK_EPI_QKNORM: qkvh / dqkvh [M, ld] (normalised values and their gradients), rn [M, 3*heads] the
 * 1/max(|.|,1e-12) factors the epilogue stored through ep1 -> dqkv [M, ldo] (raw projections), dscale[heads] += .
 * In place: dqkvh == dqkv (row stride ldo; the attention backward wrote straight into the next GEMM's operand buffer) --
 * v's gradient passes through unchanged, so only the q-hat / k-hat vectors are read and rewritten (2/3 of the vectors). */
int swiftk_qknorm_bwd(const void* qkvh, const void* dqkvh, int64_t ld, const float* rn, void* dqkv, int64_t ldo,
                      const float* scale, float* dscale, int64_t M, int heads, int head_dim, int dtype, void* stream);
/* Both in one call (round 4): dqkv [tokens, ldd] <- the gradient w.r.t. the RAW to_qkv output (the operand of the to_qkv data- and
 * weight-gradient GEMMs), dscale [heads] += d(logit scale).  Where the pipelined attention backward runs (head_dim 88) the
 * QK-norm backward is applied to the fp32 accumulators on their way out -- d(q-hat) / d(k-hat) never reach memory; elsewhere the
 * call is swiftk_window_attention_bwd_scaled followed by swiftk_qknorm_bwd in place. */
int swiftk_window_attention_bwd_qknorm(const void* qkvh, int64_t ldq, const void* o, const void* d_o, int64_t ldo, void* dqkv,
                                       int64_t ldd, const float* scale, const float* rn, float* dscale, int B, int gh, int gw,
                                       int heads, int head_dim, int shift_h, int shift_w, int dtype, void* stream);

/* Backward of the attention core (bf16, head_dim 88, PRENORM layout): dqkvh = d(q-hat | k-hat | v). */
int swiftk_window_attention_bwd(const void* qkvh, int64_t ldq, const void* o, const void* d_o, int64_t ldo, void* dqkvh, int B,
                                int gh, int gw, int heads, int head_dim, int shift_h, int shift_w, int dtype, void* stream);
/* Same, with dqkvh's own row stride ldd (>= 3*heads*head_dim) and the per-head logit scale parameter [heads] (or NULL): where exp(min(scale, ln 100)) <= 48 bounds |logit| (q-hat and
 * k-hat arrive normalised), the softmax is rebuilt without a row-maximum sweep, as in the forward kernel. */
int swiftk_window_attention_bwd_scaled(const void* qkvh, int64_t ldq, const void* o, const void* d_o, int64_t ldo, void* dqkvh,
                                       int64_t ldd, const float* scale, int B, int gh, int gw, int heads, int head_dim,
                                       int shift_h, int shift_w, int dtype, void* stream);

/* out[c] += sum_r src[r][c]  (period == 0), or out[(r % period)][c] += src[r][c]  (bias / pos_embed gradients) */
int swiftk_colsum(const float* src, int64_t lds, float* out, int64_t rows, int cols, int64_t period, void* stream);
/* The patch embedding's backward bookkeeping in one pass over d(x0) [rows = samples x period, cols] (reference: autograd of
 * `patch_embed(x) + pos_embed`, swinv2.py:309-310): bias_grad[c] += sum_r src[r][c], pos_grad[t][c] += sum_b src[b period + t][c]
 * (fixed order, no atomics), and -- dst_bf16 != NULL -- the bf16 copy [rows, ldd] with zeroed row padding that the weight-gradient
 * GEMM reads.  Replaces two swiftk_colsum calls and a swiftk_cast_pad. */
int swiftk_embed_bwd_sums(const float* src, int64_t lds, float* bias_grad, float* pos_grad, void* dst_bf16, int64_t ldd, int64_t rows,
                          int cols, int64_t period, void* stream);
/* Backward of swiftk_linear_small (pre-activation gradient dz): dx += dz W (dx zero-filled by the caller, may be NULL),
 * dW += dz^T x, dbias += sum_b dz (dW/dbias may be NULL). */
int swiftk_linear_small_bwd(const float* dz, int64_t lddz, const float* x, int64_t ldx, const float* W, int64_t ldw, float* dx,
                            int64_t lddx, float* dW, int64_t lddw, float* dbias, int B, int N, int K, void* stream);
int swiftk_silu_bwd(const float* z, const float* dy, float* dz, int64_t n, void* stream);

/*
 * Forward-mode tangent rules of the sCM pre-training loss (reference training/loss.py:186-260 runs torch.func.jvp through
 * the denoiser with jvp=True, swinv2.py:105-139).  The linear maps carry the tangent as extra GEMM rows (primal rows
 * 0..M-1, tangent rows M..2M-1 of one swiftk_gemm call); these entry points are the non-linear steps.  fp32 arithmetic,
 * `dtype` is the storage type of the activation tensors.
 */
/* d/dt of timestep_embedding (swinv2.py:44-60): demb = [cos | -sin](t w f) * w f * dt[b]. */
int swiftk_timestep_embed_jvp(const float* t, const float* dt, const float* freqs, float* demb, int B, int d,
                              float timestep_weight, void* stream);
/* y = silu(z) (y may be NULL), dy = silu'(z) dz  (LatentEmbedding, swinv2.py:67-74). */
int swiftk_silu_jvp(const float* z, const float* dz, float* y, float* dy, int64_t n, void* stream);
/* In place on qkv / dqkv [M, ld] (head layout [q88|k88|v88]): q <- q/|q| * exp(min(scale,ln100)), k <- k/|k| and their
 * tangents dq, dk (swinv2.py:123-127); v, dv untouched.  `rn` (optional, [M, 3*heads] fp32): 1/max(|.|, 1e-12) per q / k
 * vector and 1 for v -- what SWIFTK_EPI_QKNORM saves for swiftk_qknorm_bwd, so the tangent pass's primal rows can serve as
 * the saved activations of the backward pass. */
int swiftk_qknorm_jvp(void* qkv, void* dqkv, int64_t ld, const float* scale, float* rn, int64_t M, int heads, int head_dim,
                      int dtype, void* stream);
/* Explicit-softmax window attention and its tangent on pre-normalised q, k (swinv2.py:129-133 with jvp=True):
 * out = softmax(q k^T) v, dout = d/d(eps) of the same along (dq, dk, dv).  Same window/shift addressing as
 * swiftk_window_attention; head_dim 80 / 88 / 96. */
int swiftk_window_attention_jvp(const void* qkv, const void* dqkv, int64_t ldq, void* out, void* dout, int64_t ldo, int B,
                                int gh, int gw, int heads, int head_dim, int shift_h, int shift_w, int dtype, void* stream);
/* ModulatedNorm + residual and its tangent (swinv2.py:77-86): x += LN(y)(1+sc)+sh; dx += dLN(y)[dy](1+sc) + LN(y) dsc + dsh,
 * (sc|sh) = mod[b], (dsc|dsh) = dmod[b] (per sample, ld = ldmod); xT / dxT receive the new x / dx as GEMM operands. */
int swiftk_modnorm_jvp(const void* y, const void* dy, int64_t ldy, float* x, float* dx, void* xT, void* dxT, int64_t ldxT,
                       const float* gamma, const float* beta, const float* mod, const float* dmod, int64_t ldmod, int64_t M,
                       int d, int64_t rows_per_sample, float eps, int dtype, void* stream);
/* The same on the PAIR form of the stream and its tangent (bf16 operands; round 4): x and dx live as (bf16 hi, 8-bit lo) pairs,
 * hi being the GEMM operands xT / dxT -- 16 bytes per element instead of 24.  hi is read from (xT_in, dxT_in) and written to
 * (xT, dxT), which may be the same buffers; x_lo / dx_lo [M, d] bytes, updated in place (swiftk_split_pair creates them). */
int swiftk_modnorm_jvp_pair(const void* y, const void* dy, int64_t ldy, const void* xT_in, const void* dxT_in, void* xT, void* dxT,
                            int64_t ldxT, void* x_lo, void* dx_lo, const float* gamma, const float* beta, const float* mod,
                            const float* dmod, int64_t ldmod, int64_t M, int d, int64_t rows_per_sample, float eps, void* stream);
/* A linear map of the tangent pass WITH the non-linear step behind it, in one launch (round 4): A [2 Mh, lda] bf16 holds the
 * primal rows 0..Mh-1 and the tangent rows Mh..2Mh-1 (Mh % 128 == 0).  An output tile of the persistent GEMM takes 128 primal rows
 * and THEIR 128 tangent rows, laid out so that a lane's accumulators hold a primal element and its tangent: the tangent rules
 * below run on the fp32 accumulators, and neither the raw products nor their tangents reach memory.
 *   SWIFTK_EPI_QKNORM_JVP  (to_qkv, swinv2.py:121-127): C [2 Mh, ldc] bf16 <- (q-hat | k-hat | v) and, in the tangent rows, their
 *       tangents -- what swiftk_gemm + swiftk_qknorm_jvp leave; `scale` [heads]; `rn` (optional, [Mh, N / head_dim] fp32) as
 *       swiftk_qknorm_jvp writes it; C2 unused.
 *   SWIFTK_EPI_SWIGLU_JVP  (w1, swinv2.py:99-100; W rows interleaved gate / up): C2 [2 Mh, ldc2] bf16 <- silu(gate) * up in the primal
 *       rows and its tangent in the tangent rows -- what swiftk_gemm + swiftk_swiglu_jvp leave; C (optional, [Mh, ldc] bf16) <- the
 *       primal pre-activations (the saved activation of SWIFTK_EPI_SWIGLU_BWD). */
int swiftk_gemm_jvp(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t Mh, int64_t N, int64_t K,
                    int epilogue, const float* scale, float* rn, int head_dim, void* C2, int64_t ldc2, void* stream);
/* SwiGLU and its tangent on interleaved (gate_j, up_j) columns (swinv2.py:99-100). */
int swiftk_swiglu_jvp(const void* h, const void* dh, int64_t ldh, void* out, void* dout, int64_t ldo, int64_t M, int mlp,
                      int dtype, void* stream);
/* sCM regression target (loss.py:236-247): g = -cos^2 t (sd F - dxt) - r (cos t sin t x_t + sd dF), g /= (rms_b(g) + 0.1),
 * target = F + g; x_t = xt_over_sd * sd.  ss_scratch: B floats. */
int swiftk_scm_target(const float* F, const float* dxt, const float* xt_over_sd, const float* dF, const float* t, float r,
                      float sigma_data, float* target, float* ss_scratch, int B, int64_t per_sample, void* stream);

/* Validation RMSE sums (training/validate.py:96-107): sq[0] += sum (y - t)^2 over everything, sq[1 + c] += sum_{b,h,w}
 * w_lat[h] (y - t)^2 per channel; y [B,C,H,W] contiguous, t the same with batch stride t_batch_stride (a [B, days, C, H, W]
 * slice).  The caller zero-fills sq (1 + C floats), divides by the counts and takes the roots. */
int swiftk_rmse_sums(const float* y, const float* t, int64_t t_batch_stride, const float* w_lat, float* sq, int B, int C, int H,
                     int W, void* stream);
/* Ensemble evaluation sums (eval/metrics.py:39-134), pred [B, N, V, H, W], y [B, V, H, W], out [B, V, 4] (zero-filled by
 * the caller): per (sample, variable) the latitude-weighted grid sums of (ens-mean - y)^2, sum_n |x_n - y|,
 * sum_{n,n'} |x_n - x_n'| and the unbiased member variance; 2 <= N <= 64.  RMSE / CRPS / spread-skill follow on the host. */
int swiftk_ensemble_sums(const float* pred, const float* y, const float* w_lat, float* out, int B, int N, int V, int H, int W,
                         void* stream);
/* Almost-fair CRPS over m members (loss.py:343-371,445): *loss += 1/(B H W) sum w_var[c] w_lat[h] crps; dpreds optional. */
int swiftk_crps_loss(const float* preds, const float* target, const float* w_var, const float* w_lat, float* loss,
                     float* dpreds, int m, int B, int C, int H, int W, float alpha, float gscale, void* stream);
/* TrigFlow (loss.py:132-160): x_t/sigma_d and v_t from (x, z ~ N(0,1), t); then the weighted loss and its gradients. */
int swiftk_trigflow_prep(const float* x, const float* z, const float* t, float* xt_over_sd, float* vt, float sigma_data,
                         int B, int64_t per_sample, void* stream);
int swiftk_trigflow_loss(const float* F, const float* vt, const float* logvar, const float* w_var, const float* w_lat,
                         float* loss, float* dF, float* dlogvar, float sigma_data, int B, int C, int H, int W, float gscale,
                         void* stream);
/* out = a[b]*x + c[b]*y (y may be NULL) ;  out = x + coef[channel]*y (x may be NULL) */
int swiftk_axpby_per_sample(float* out, const float* a, const float* x, const float* c, const float* y, int B,
                            int64_t per_sample, void* stream);
int swiftk_channel_axpy(float* out, const float* x, const float* y, const float* coef, int B, int C, int64_t hw,
                        void* stream);

/* fp32 -> three bf16 column blocks of `cols` columns each (cols % 4 == 0, ldd >= 3*cols, the rest of a row zero):
 * hi = bf16(v), lo = bf16(v - hi).  order 0: [hi | lo | hi] (activations), order 1: [hi | hi | lo] (weights) -- the operand
 * layout of the SWIFTK_BF16X3 engine: swiftk_gemm over K' = 3*cols on these is hi hi' + lo hi' + hi lo' in fp32 accumulators,
 * an fp32-grade product (measured 4.5e-6 relative against fp64) at the bf16 MFMA rate / 3. */
int swiftk_split3(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int order, void* stream);

/* fp32 -> dtype copy with row padding: dst[r][c] = src[r][c] for c < cols, 0 for cols <= c < ldd. */
int swiftk_cast_pad(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype,
                    void* stream);

/* A trainable weight's two bf16 GEMM operands from its fp32 master copy in one pass (the training engine refreshes them after every
 * optimizer step): out [rows, ldo] = bf16(W) with zeroed row padding -- the forward operand -- and out_t [cols, ldt] = bf16(W)^T with
 * zeroed row padding -- the "W" of the data-gradient GEMM dX = dY W.  interleave = n > 0 (rows == 2 n): output row 2 j + s takes
 * input row s n + j, the (gate, up) interleave SWIFTK_EPI_SWIGLU wants of FeedForward.w1 (swinv2.py:96-101). */
int swiftk_cast_pad_t(const float* W, int64_t ldw, int64_t rows, int64_t cols, void* out, int64_t ldo, void* out_t, int64_t ldt,
                      int64_t interleave, void* stream);

/*
 * to_qkv + cosine norm + shifted-window attention of one layer in one kernel (bf16; head_dim 80 / 88 / 96 = the 468 M variant,
 * Swift-B, the 664 M variant of configs/experiment/era5-swinv2-1.4-scm.yaml:21-36; 16 x 16 windows; K >= 128): the q / k / v
 * slab of a (sample, window, head) is produced, normalised and consumed on the CU; only the attention output reaches memory.
 * Replaces src/swift/models/swinv2.py:119-136 (to_qkv, split, normalise, scale, attention) and :185-208 (roll,
 * window_partition, window_reverse) -- i.e. swiftk_gemm_qkv_tiled + swiftk_window_attention.
 *   x    [B*gh*gw, ldx] bf16 token-major (K valid columns; K = 16.5 k-tiles style padding as in swiftk_gemm)
 *   w    [3*heads*head_dim, ldw] bf16 (to_qkv.weight as stored: per-head [q|k|v] rows)
 *   out  [B*gh*gw, ldo] bf16, head h in columns [head_dim h, head_dim (h + 1)), token order (un-rolled)
 * SWIFTK_ESHAPE for any other head_dim, a grid that is not a multiple of the window, a shift outside the grid.
 */
int swiftk_qkv_attention_fused(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* scale, void* out,
                               int64_t ldo, int64_t K, int B, int gh, int gw, int heads, int head_dim, int shift_h,
                               int shift_w, void* stream);

/* ------------------------------------------------------------------------ *
 * Whole-network forward (the operator boundary itself).
 * ------------------------------------------------------------------------ */

typedef struct swiftk_layer {
    const void* qkv_w;   /* [3*heads*hd, kd] dtype                     (to_qkv.weight)            */
    const void* wo_w;    /* [d, kd]          dtype                     (wo.weight)                */
    const void* w1_w;    /* [2*mlp, kd]      dtype, rows interleaved gate_j, up_j (w1.weight)      */
    const void* w2_w;    /* [d, kmlp]        dtype                     (w2.weight)                */
    const float* scale;  /* [heads]                                    (Attention.scale)          */
    const float* ln1_g;  /* [d] attn norm.norm.weight */
    const float* ln1_b;
    const float* ln2_g;  /* [d] ff norm.norm.weight   */
    const float* ln2_b;
    const void* qkv_w_f32;   /* SWIFTK_BF16X3 with x3_exact bit 6 (adaptive to_qkv), else NULL: to_qkv.weight as fp32 operands
                                [3*heads*hd, kd] beside the split form in qkv_w                                              */
    int32_t qk_exact_pairs;  /* ... and which PAIRS of heads (bits 0..15: bit p = heads 2p, 2p+1) are recomputed on the exact-fp32 kernel:
                                those whose logit scale exp(min(scale, ln 100)) exceeds the packer's threshold -- the split
                                product's 4.5e-6 reaches the softmax multiplied by that scale.  Bits 16..31 (optional, models of up
                                to 16 heads): the hot HEADS themselves (bit 16 + h); when given, each is recomputed alone (one tile
                                column per hot head instead of two per pair) and the pair's other head keeps the split product   */
} swiftk_layer;

typedef struct swiftk_model {
    int32_t dtype;                 /* SWIFTK_F32 | SWIFTK_BF16 | SWIFTK_BF16X3 (fp32 activations and k-paddings as for SWIFTK_F32;
                                      GEMM weights stored by swiftk_split3(order 1) with row stride
                                      swiftk_gemm_k_pad(SWIFTK_BF16, 3*K), except those whose GEMM stays on the
                                      exact-fp32 kernel -- the x3_exact mask below --
                                      which are fp32 operands as for SWIFTK_F32) */
    int32_t H, W, p1, p2;          /* image and patch size      */
    int32_t in_ch, out_ch;         /* 141, 69                   */
    int32_t depth, dim, heads;     /* 12, 1056, 12              */
    int32_t mlp;                   /* int(8/3 dim) = 2816       */
    int32_t wh, ww, sh, sw;        /* window 16x16, shift 8x8   */
    int32_t aux_dim;
    int32_t has_logvar;
    int32_t x3_exact;              /* SWIFTK_BF16X3 only: GEMMs whose weights were packed as fp32 operands (bit 0 to_qkv, 1 wo,
                                      2 w1, 3 w2, 4 patch embed, 5 head; bit 6: to_qkv split EXCEPT the head pairs of
                                      swiftk_layer.qk_exact_pairs); a property of THIS model's weight buffers, set by
                                      whoever packed them (swiftk_get_tuning(11) is the library's default: 17) */
    float timestep_weight;
    int64_t kd;                    /* swiftk_gemm_k_pad(dtype, dim)            */
    int64_t kmlp;                  /* swiftk_gemm_k_pad(dtype, mlp)            */
    int64_t kpe;                   /* swiftk_gemm_k_pad(dtype, in_ch*p1*p2)    */
    const void* pe_w;              /* [d, kpe] dtype   (patch_embed.emb.weight) */
    const float* pe_b;             /* [d]                                       */
    const float* pos;              /* [gh*gw, d]       (pos_embed)              */
    const float* freqs;            /* [d/2]                                     */
    const float* aux_w;            /* [d, aux_dim]     (auxiliary_embed)        */
    const float* aux_b;
    const float* l1_w;             /* [d, d] fp32      (latent_embed.l1)        */
    const float* l1_b;
    const float* l2_w;
    const float* l2_b;
    const float* mod_w;            /* [depth*2*2d, d] fp32: layer i attn at rows (2i)*2d, ff at (2i+1)*2d */
    const float* mod_b;            /* [depth*2*2d]                              */
    const float* logvar_w;         /* [1, d] or NULL                            */
    const float* logvar_b;
    const void* head_w;            /* [round_up(out_ch*p1*p2, 4), kd] dtype (head.head.0.weight, zero rows appended) */
    const swiftk_layer* layers_host; /* HOST array of `depth` entries           */
} swiftk_model;

/*
 * The optimisation step of the training loop (trainer.py:219-247) in one pass: nan_to_num of the (all-reduced) gradients,
 * the torch.optim.Adam / AdamW update (torch/optim/adam.py single-tensor rule) and the EMA rule
 * p_ema <- p_net.lerp(p_ema, ema_beta).  `chunks` is a DEVICE table, one entry per <= 16384 consecutive elements of one
 * parameter tensor; gradients and both moments are flat fp32 buffers indexed by flat_off.  `hyper_host` is read on the host.
 */
#define SWIFTK_OPT_MAX_GROUPS 8
typedef struct {
    float* p;          /* parameter elements of this chunk (fp32)                  */
    float* ema;        /* the EMA copy's elements, or NULL                         */
    int64_t flat_off;  /* offset of the chunk in grad / exp_avg / exp_avg_sq        */
    int32_t n;         /* elements in the chunk                                    */
    int32_t group;     /* optimizer.param_groups index (lr / weight_decay)         */
} swiftk_opt_chunk;
typedef struct {
    float lr[SWIFTK_OPT_MAX_GROUPS];
    float weight_decay[SWIFTK_OPT_MAX_GROUPS];
    float step_size[SWIFTK_OPT_MAX_GROUPS]; /* lr / (1 - beta1^t)                      */
    float beta1, beta2, eps;
    float bias2_sqrt;                       /* sqrt(1 - beta2^t)                        */
    float ema_beta;
    int32_t decoupled;                      /* 1: AdamW (p *= 1 - lr wd), 0: Adam (g += wd p) */
} swiftk_opt_hyper;
int swiftk_adamw_ema_step(const swiftk_opt_chunk* chunks, int n_chunks, float* grad_flat, float* exp_avg_flat,
                          float* exp_avg_sq_flat, const swiftk_opt_hyper* hyper_host, void* stream);

/* Bytes of scratch swiftk_swinv2_forward needs for batch B (0 on a bad model). */
int64_t swiftk_workspace_bytes(const swiftk_model* m, int B);

/*
 * out = alpha * xt + beta * SwinV2(cat[src0*s0, src1*s1, src2*s2], t, aux)
 * i.e. swinv2.py:305-330 with precond.py:139-148's concat folded into the
 * patch gather and (optionally) diffusion.py:459's update folded into the
 * un-patchify.  Pass xt = NULL for the bare network output.
 *   src_k NCHW fp32, channels c0 + c1 + c2 == in_ch (unused sources NULL / 0)
 *   t [B], aux [B, aux_dim] (or NULL), alpha/beta [B] fp32 (NULL -> 0 / 1)
 *   out [B, out_ch, H, W] fp32;  logvar [B] fp32 or NULL
 */
int swiftk_swinv2_forward(const swiftk_model* m, const float* src0, int c0, float s0, const float* src1, int c1, float s1,
                          const float* src2, int c2, float s2, const float* t, const float* aux, const float* xt,
                          const float* alpha, const float* beta, float* out, float* logvar, int B, void* workspace,
                          int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWIFTK_H */
