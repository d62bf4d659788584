"""-m gpu: every HIP kernel, called through the C ABI, against the CPU oracle on the same inputs.

Tolerances (relative L2 unless noted):
  fp32 kernels   1e-5 per op (fp32 MFMA is an exact fp32 FMA chain; only summation order differs)
  bf16 kernels   operands rounded to bf16 (2^-9 relative), fp32 accumulation: 6e-3 per GEMM-like op
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu

F32_TOL = 1e-5
BF16_TOL = 6e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def rnd(shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


def to_dt(x, dt, dev):
    return x.to(dev).to(dt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(300, 704, 192), (2048, 3168, 1088), (512, 276, 1088), (77, 36, 64)])
def test_gemm_plain(dev, dt, shape):
    from swift_amd import ops
    M, N, K = shape
    K = ops.k_pad(dt, K)
    a, w = rnd((M, K), 1), rnd((N, K), 2, 0.05)
    ad, wd = to_dt(a, dt, dev), to_dt(w, dt, dev)
    c = ops.gemm(ad, wd)
    ref = ad.float().cpu().double() @ wd.float().cpu().double().T  # same rounded operands, exact accumulate
    assert c.dtype == dt
    assert rel_l2(c.float().cpu(), ref) < (F32_TOL if dt == torch.float32 else 4e-3)
    c32 = ops.gemm(ad, wd, out_dtype=torch.float32)
    assert rel_l2(c32.cpu(), ref) < F32_TOL


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_strided_rows_and_identity(dev, dt):
    """A = I picks rows of an ASYMMETRIC W: catches transposed / permuted fragment maps exactly."""
    from swift_amd import ops
    K = 128
    a = torch.zeros(256, 192)
    a[:, :K] = torch.eye(256)[:, :K]
    a[:, K:] = float("nan")  # beyond lda-visible K: must never be read
    w = (torch.arange(352 * K, dtype=torch.float32).reshape(352, K) % 251) - 125.0
    ad, wd = to_dt(a, dt, dev), to_dt(w, dt, dev)
    c = ops.gemm(ad[:, :K], wd, out_dtype=torch.float32)
    ref = torch.zeros(256, 352)
    ref[:K] = w.T
    assert torch.equal(c.cpu(), ref)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_bias_pos(dev, dt):
    from swift_amd import ops
    M, N, K = 1024, 1056, ops.k_pad(dt, 564)
    a, w = rnd((M, K), 3), rnd((N, K), 4, 0.05)
    bias, pos = rnd((N,), 5), rnd((256, N), 6)
    ad, wd = to_dt(a, dt, dev), to_dt(w, dt, dev)
    c = ops.gemm(ad, wd, out_dtype=torch.float32, epilogue=ops.EPI_BIAS_POS, bias=bias.to(dev), pos=pos.to(dev))
    ref = ad.float().cpu().double() @ wd.float().cpu().double().T + bias.double() + pos.double().repeat(4, 1)
    assert rel_l2(c.cpu(), ref) < F32_TOL


@pytest.mark.parametrize("H", [2816, 2560, 3072])
def test_gemm_swiglu_both_outputs(dev, H):
    """EPI_SWIGLU_BOTH (training forward): the pre-activation and silu(gate) * up from one launch, each bit-equal to the
    launch that produces it alone (plain GEMM / EPI_SWIGLU); ragged last row tile; all three tile widths."""
    from swift_amd import _lib, ops
    M, K = 1000, ops.k_pad(torch.bfloat16, 1056)
    a, w = to_dt(rnd((M, K), 7), torch.bfloat16, dev), to_dt(rnd((2 * H, K), 8, 0.03), torch.bfloat16, dev)
    h = torch.full((M, 2 * H), float("nan"), dtype=torch.bfloat16, device=dev)
    hm = torch.full((M, H + 64), float("nan"), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, h.data_ptr(), 2 * H, M, 2 * H, K, _lib.BF16, _lib.BF16,
                                      _lib.EPI_SWIGLU_BOTH, None, hm.data_ptr(), H + 64, torch.cuda.current_stream().cuda_stream), "gemm")
    assert torch.equal(h.view(torch.int16), ops.gemm(a, w).view(torch.int16))
    assert torch.equal(hm[:, :H].view(torch.int16), ops.gemm(a, w, epilogue=ops.EPI_SWIGLU).view(torch.int16))
    assert bool(hm[:, H:].isnan().all())  # nothing written past the H valid columns


@pytest.mark.parametrize("H", [2816, 2560, 3072])
def test_gemm_swiglu_backward_epilogue(dev, H):
    """EPI_SWIGLU_BWD: d(pre-activation) from dY @ W2 and the saved (gate, up), against autograd of silu(gate) * up applied
    to the fp64 product (the fused form skips the bf16 rounding of d(hidden) the two-kernel path has)."""
    from swift_amd import _lib, ops
    M, K = 1000, ops.k_pad(torch.bfloat16, 1056)
    a, w = to_dt(rnd((M, K), 17), torch.bfloat16, dev), to_dt(rnd((H, K), 18, 0.03), torch.bfloat16, dev)
    h = to_dt(rnd((M, 2 * H), 19), torch.bfloat16, dev)
    dh = torch.full((M, 2 * H + 64), float("nan"), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, dh.data_ptr(), 2 * H + 64, M, H, K, _lib.BF16, _lib.BF16,
                                      _lib.EPI_SWIGLU_BWD, None, h.data_ptr(), 2 * H, torch.cuda.current_stream().cuda_stream), "gemm")
    dhid = a.float().cpu().double() @ w.float().cpu().double().T
    hc = h.float().cpu().double().requires_grad_(True)
    (torch.nn.functional.silu(hc[:, 0::2]) * hc[:, 1::2]).backward(dhid)
    assert rel_l2(dh[:, :2 * H].float().cpu(), hc.grad) < 4e-3
    assert bool(dh[:, 2 * H:].isnan().all())


@pytest.mark.parametrize("K", [1056, 2816, 576])
def test_gemm_fp32_two_level_accumulation(dev, K):
    """swiftk_gemm_chunked (round 4): fp32 operands, MFMA chains of 256 k met through a scratch slab.  The plain kernel's
    single chain over K ends measurably further from the fp64 product than the chained form (tests/fp32_bisect.py traced the
    exact engine's 1.8 x excess over the CPU's fp32 error to it); both stay inside the per-op tolerance, the epilogues see
    the same accumulators."""
    from swift_amd import ops
    M, N = 2048 + 24, 704
    a, w = rnd((M, K), 41), rnd((N, K), 42, 0.05)
    ad, wd = a.to(dev), w.to(dev)
    ref = a.double() @ w.double().T
    plain, chained = ops.gemm(ad, wd), ops.gemm_chunked(ad, wd, chunk_k=256)
    e_p, e_c = rel_l2(plain.cpu(), ref), rel_l2(chained.cpu(), ref)
    e_cpu = rel_l2(a @ w.T, ref)
    print(f"fp32 GEMM K={K}: rel-L2 vs fp64: one chain {e_p:.2e}, chains of 256 {e_c:.2e}, CPU sgemm {e_cpu:.2e}")
    assert e_p < F32_TOL and e_c < F32_TOL
    assert e_c < (0.8 * e_p if K > 1000 else 1.05 * e_p)
    assert rel_l2(chained.cpu(), plain.cpu()) < 2e-6
    # several tiles per workgroup (the parked slab is re-used tile after tile) and a ragged last row tile
    M2 = 256 * 300 + 40
    a2 = rnd((M2, 192), 43).to(dev)
    w2 = rnd((352, 192), 44, 0.05).to(dev)
    assert rel_l2(ops.gemm_chunked(a2, w2, chunk_k=64).cpu(), a2.cpu().double() @ w2.cpu().double().T) < F32_TOL
    # fused epilogues take the merged accumulators
    H = 352
    w1 = rnd((2 * H, K), 45, 0.03)
    wi = w1.view(2, H, K).permute(1, 0, 2).reshape(2 * H, K).contiguous()
    c = ops.gemm_chunked(ad, wi.to(dev), epilogue=ops.EPI_SWIGLU)
    h = a.double() @ w1.double().T
    assert rel_l2(c.cpu(), torch.nn.functional.silu(h[:, :H]) * h[:, H:]) < F32_TOL


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [1056, 1280, 1536])
def test_gemm_accumulate(dev, dt, N):
    """EPI_ACCUM: C += A W^T in fp32 (the backward pass's `dx += dY W`), all three tile widths, a ragged last row tile."""
    from swift_amd import ops
    if dt == torch.float32 and N != 1056:
        pytest.skip("fp32 operands use the 352-wide tile only")
    M, K = 1000, ops.k_pad(dt, 704)
    a, w, c0 = rnd((M, K), 11), rnd((N, K), 12, 0.05), rnd((M, N), 13)
    ad, wd = to_dt(a, dt, dev), to_dt(w, dt, dev)
    c = c0.to(dev)
    assert ops.gemm(ad, wd, out=c, epilogue=ops.EPI_ACCUM) is c
    ref = c0.double() + ad.float().cpu().double() @ wd.float().cpu().double().T
    assert rel_l2(c.cpu(), ref) < F32_TOL
    with pytest.raises(Exception):
        ops.gemm(ad, wd, out=torch.zeros(M, N, dtype=dt, device=dev) if dt != torch.float32 else None, epilogue=ops.EPI_ACCUM,
                 out_dtype=torch.bfloat16)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_swiglu(dev, dt):
    from swift_amd import ops
    M, H, K = 512, 2816, ops.k_pad(dt, 1056)
    a, w1 = rnd((M, K), 7), rnd((2 * H, K), 8, 0.03)
    wi = w1.view(2, H, K).permute(1, 0, 2).reshape(2 * H, K)  # gate_j, up_j interleaved
    ad, wd = to_dt(a, dt, dev), to_dt(wi, dt, dev)
    c = ops.gemm(ad, wd, epilogue=ops.EPI_SWIGLU)
    assert c.shape == (M, H)
    h = ad.float().cpu().double() @ to_dt(w1, dt, dev).float().cpu().double().T
    ref = torch.nn.functional.silu(h[:, :H]) * h[:, H:]
    assert rel_l2(c.float().cpu(), ref) < (F32_TOL if dt == torch.float32 else 4e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shift", [(0, 0), (8, 8), (3, 5)])
def test_window_attention(dev, dt, shift):
    from oracle.swinv2 import cosine_window_attention, window_token_index
    from swift_amd import ops
    B, grid, heads, hd = 2, (32, 48), 12, 88
    n = grid[0] * grid[1]
    qkv = rnd((B, n, 3 * heads * hd), 9)
    scale = torch.log(torch.tensor([10.0, 3.0, 30.0, 200.0, 1.0, 10.0, 50.0, 99.0, 101.0, 5.0, 20.0, 10.0]))
    qd = to_dt(qkv, dt, dev)
    out = ops.window_attention(qd, scale.to(dev), grid, heads, shift)
    idx = window_token_index(grid, (16, 16), shift)
    src = qd.float().cpu()
    ow = cosine_window_attention(src[:, idx.reshape(-1)].reshape(B * idx.shape[0], 256, -1), scale.view(1, heads, 1, 1),
                                 heads, naive=True)
    ref = torch.empty(B, n, heads * hd)
    ref[:, idx.reshape(-1)] = ow.reshape(B, n, -1)
    assert rel_l2(out.float().cpu(), ref) < (2e-5 if dt == torch.float32 else 1.2e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_window_attention_sharp_rows(dev, dt):
    """Logits up to +-100 (scale clamp) with near-duplicate keys: exercises the row-max subtraction."""
    from oracle.swinv2 import cosine_window_attention
    from swift_amd import ops
    heads, hd = 12, 88
    qkv = rnd((1, 256, 3 * heads * hd), 10)
    v = qkv.view(1, 256, heads, 3, hd)
    v[:, :, :, 1] = v[:, :, :, 0] * 3.0 + 0.01 * rnd((1, 256, heads, hd), 11)  # k ~ parallel to q -> cos ~ 1
    scale = torch.full((heads,), 9.0)  # clamped to ln(100)
    qd = to_dt(qkv, dt, dev)
    out = ops.window_attention(qd, scale.to(dev), (16, 16), heads)
    ref = cosine_window_attention(qd.float().cpu(), scale.view(1, heads, 1, 1), heads, naive=True)
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out.float().cpu(), ref) < (5e-5 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rps", [200, 208])  # 208 = 13 x 16: the chunked kernel (16-row chunks inside one sample); 200: row per wave
def test_modnorm_residual(dev, dt, rps):
    from swift_amd import ops
    B, d = 3, 1056
    M = B * rps
    y, x = rnd((M, d), 12, 2.0) + 0.5, rnd((M, d), 13)
    gamma, beta, mod = 1 + 0.1 * rnd((d,), 14), 0.1 * rnd((d,), 15), 0.3 * rnd((B, 5 * 2 * d), 16)
    yd = to_dt(y, dt, dev)
    xd = x.to(dev).clone()
    ld = ops.k_pad(dt, d)
    xc = torch.full((M, ld), 7.0, dtype=dt, device=dev)
    msl = mod.to(dev)[:, 4 * d: 6 * d]  # a strided slice, like layer 2 of the concatenated modulation
    ops.modnorm_residual(yd, xd, gamma.to(dev), beta.to(dev), msl, rps, xcopy=xc)
    yn = torch.nn.functional.layer_norm(yd.float().cpu(), (d,), gamma, beta, 1e-6)
    m = mod[:, 4 * d: 6 * d].repeat_interleave(rps, 0)
    ref = x + yn * (1 + m[:, :d]) + m[:, d:]
    assert rel_l2(xd.cpu(), ref) < F32_TOL
    assert rel_l2(xc[:, :d].float().cpu(), ref) < (F32_TOL if dt == torch.float32 else 3e-3)
    assert (xc[:, d:].float() == 7.0).all()  # pad columns untouched


@pytest.mark.parametrize("lo_bits", [8, 16])
@pytest.mark.parametrize("rps", [512, 48])
def test_modnorm_residual_pair(dev, rps, lo_bits):
    """The bf16 engine's residual stream as a pair -- hi = bf16(x) plus a bf16 or an 8-bit low part (swiftk_split_pair,
    swiftk_modnorm_residual_pair): the split is exact bf16 arithmetic (bit-equal to torch), the update is the oracle's
    ModulatedNorm + residual (swinv2.py:83-86, 211-212) on the value the pair stands for, stored back to 2^-17 relative, and hi
    is what a bf16 cast of the new x gives."""
    from oracle.swinv2 import modulated_norm
    from swift_amd import ops
    B, d = 3, 1056
    M = B * rps
    y, x = rnd((M, d), 12, 2.0) + 0.5, rnd((M, d), 13)
    p = {"n.norm.weight": 1 + 0.1 * rnd((d,), 14), "n.norm.bias": 0.1 * rnd((d,), 15),
         "n.modulation.weight": 0.02 * rnd((2 * d, d), 16), "n.modulation.bias": 0.1 * rnd((2 * d,), 17)}
    lat = rnd((B, d), 18)
    ld = ops.k_pad(torch.bfloat16, d)
    hi, lo = ops.split_pair(x.to(dev), ld, lo_bits)
    xh = x.bfloat16()
    assert torch.equal(hi[:, :d].cpu(), xh) and (hi[:, d:].float() == 0).all()
    if lo_bits == 16:
        assert torch.equal(lo.cpu(), (x - xh.float()).bfloat16())
    else:
        assert lo.dtype == torch.uint8
    x_in = ops.pair_value(hi, lo, d).cpu()
    assert float(((x_in - x).abs() / x.abs().clamp_min(1e-20)).max()) < 2.0 ** -15
    mod = torch.nn.functional.linear(lat, p["n.modulation.weight"], p["n.modulation.bias"])  # [B, 2d]: scale | shift
    yd = to_dt(y, torch.bfloat16, dev)
    hi[:, d:] = 7.0
    if lo_bits == 8:
        # d = 1056 with the 8-bit low part runs the PACKED kernel (four rows per wave pass, no idle lanes); the row-per-wave
        # form of the same arithmetic (tuning key 6 bit 2) must agree with it bit for bit
        from swift_amd import _lib
        h2, l2 = hi.clone(), lo.clone()
        _lib.lib().swiftk_set_tuning(6, 7)
        ops.modnorm_residual_pair(yd, h2, l2, p["n.norm.weight"].to(dev), p["n.norm.bias"].to(dev), mod.to(dev), rps, d)
        _lib.lib().swiftk_set_tuning(6, 3)
    ops.modnorm_residual_pair(yd, hi, lo, p["n.norm.weight"].to(dev), p["n.norm.bias"].to(dev), mod.to(dev), rps, d)
    if lo_bits == 8:
        # (same formulas; the row sums associate differently, so a few last bits of the statistics may differ)
        assert float((h2[:, :d] != hi[:, :d]).float().mean()) < 1e-3
        assert rel_l2(ops.pair_value(h2, l2, d).cpu(), ops.pair_value(hi, lo, d).cpu()) < 2e-6
    ref = x_in.view(B, rps, d) + modulated_norm(yd.float().cpu().view(B, rps, d), lat, p, "n.")
    got = ops.pair_value(hi, lo, d).cpu()
    e = rel_l2(got, ref.view(M, d))
    print(f"pair ModulatedNorm (rows per sample {rps}, {lo_bits}-bit low part): pair value vs oracle rel-L2 {e:.2e}")
    assert e < 1e-5
    assert float((got - ref.view(M, d)).abs().max() / ref.abs().max()) < 2.0 ** -15
    # hi is the bf16 operand of the new x: lo stays within half an ulp of hi; re-rounding hi + lo differs only where lo rounded
    # up to exactly half an ulp (a tie, ~2^-9 of the elements, half of which resolve the other way)
    assert float((hi[:, :d].cpu() != got.bfloat16()).float().mean()) < 3e-3
    if lo_bits == 16:
        assert float((lo.float().abs().cpu() > 2.0 ** -8 * hi[:, :d].float().abs().cpu() + 1e-30).float().mean()) == 0.0
    assert (hi[:, d:].float() == 7.0).all()                  # pad columns untouched
    # against the fp32-stream kernel on the same inputs: the operand copies agree except where x sits within 2^-17 of a tie
    xd, xc = x_in.to(dev).clone(), torch.zeros(M, ld, dtype=torch.bfloat16, device=dev)
    ops.modnorm_residual(yd, xd, p["n.norm.weight"].to(dev), p["n.norm.bias"].to(dev), mod.to(dev), rps, xcopy=xc)
    assert float((xc[:, :d] != hi[:, :d]).float().mean()) < 1e-3
    assert rel_l2(got, xd.cpu()) < 1e-5


def test_unit_noise_vs_oracle(dev):
    """swiftk_unit_noise: the generator's words bit for bit against the numpy Philox4x32-10 of oracle/noise.py (which
    tests/test_oracle_golden.py pins to Random123's known answers), the normals against its Box-Muller, and the properties a
    rollout relies on: a pure function of (seed, step, element) whatever the batch, moments of N(0, 1)."""
    from oracle import noise as onoise
    from swift_amd import ops
    n = 69 * 32 * 64
    seeds = [12345, (7 << 40) + 99, 2 ** 63 - 1, 0]
    sd = torch.tensor(seeds, dtype=torch.int64, device=dev)
    out = torch.empty(len(seeds), n, device=dev)
    for step in (0, 3, (1 << 33) + 5):
        ops.unit_noise(out, sd, step, raw=True)
        bits = out.view(torch.int32).cpu().numpy().view(np.uint32)
        for b, s_ in enumerate(seeds):
            assert np.array_equal(bits[b], onoise.unit_bits(s_, step, n)), (b, step)
        ops.unit_noise(out, sd, step)
        z = out.cpu().numpy()
        for b, s_ in enumerate(seeds):
            ref = onoise.unit_noise(s_, step, n)
            assert np.abs(z[b] - ref).max() < 2e-5, (b, step, np.abs(z[b] - ref).max())
    # device-side step counter (what a captured step reads) == host-side step; one unit alone == the same unit in a batch
    ctr = torch.tensor([3], dtype=torch.int64, device=dev)
    a = ops.unit_noise(torch.empty(len(seeds), n, device=dev), sd, 0, step_dev=ctr).clone()
    ops.unit_noise(out, sd, 3)
    assert torch.equal(a, out)
    ops.counter_add(ctr, 2)
    assert int(ctr.item()) == 5
    one = ops.unit_noise(torch.empty(1, n, device=dev), sd[2:3], 3)
    assert torch.equal(one[0], out[2])
    big = ops.unit_noise(torch.empty(2, 69 * 128 * 256, device=dev), sd[:2], 11)
    assert abs(float(big.mean())) < 2e-3 and abs(float(big.var()) - 1.0) < 3e-3
    assert abs(float((big[0] * big[1]).mean())) < 3e-3 and abs(float((big[0, :-1] * big[0, 1:]).mean())) < 3e-3
    assert float(big.abs().max()) < 6.5 and torch.isfinite(big).all()


@pytest.mark.parametrize("lo_bits", [8, 16])
def test_modnorm_residual_pair_from_splitk_slabs(dev, lo_bits):
    """Round 4, one unit per step: wo / w2 run as two k-ranges into fp32 slabs (swiftk_gemm_splitk) and the pair-form norm kernel
    sums them on its way (swiftk_modnorm_residual_pair_slabs).  Slabs from the real split-K GEMM; against the oracle's
    ModulatedNorm on the fp64 product, and against the one-launch GEMM + bf16 y path (which rounds y to bf16 first)."""
    from oracle.swinv2 import modulated_norm
    from swift_amd import _lib, ops
    B, rps, d, K = 2, 256, 1056, 2816
    M = B * rps
    a, w, x = rnd((M, K), 51), rnd((d, K), 52, 0.03), rnd((M, d), 53)
    ad, wd = to_dt(a, torch.bfloat16, dev), to_dt(w, torch.bfloat16, dev)
    slabs = torch.full((2, M, d), float("nan"), device=dev)
    _lib.check(_lib.lib().swiftk_gemm_splitk(ad.data_ptr(), K, wd.data_ptr(), K, slabs.data_ptr(), d, M * d, M, d, K, _lib.BF16, 2,
                                             torch.cuda.current_stream().cuda_stream), "swiftk_gemm_splitk")
    y64 = ad.float().cpu().double() @ wd.float().cpu().double().T
    assert rel_l2((slabs[0] + slabs[1]).cpu(), y64) < F32_TOL
    p = {"n.norm.weight": 1 + 0.1 * rnd((d,), 14), "n.norm.bias": 0.1 * rnd((d,), 15),
         "n.modulation.weight": 0.02 * rnd((2 * d, d), 16), "n.modulation.bias": 0.1 * rnd((2 * d,), 17)}
    lat = rnd((B, d), 18)
    mod = torch.nn.functional.linear(lat, p["n.modulation.weight"], p["n.modulation.bias"]).to(dev)
    ld = ops.k_pad(torch.bfloat16, d)
    hi, lo = ops.split_pair(x.to(dev), ld, lo_bits)
    x_in = ops.pair_value(hi, lo, d).cpu()
    ops.modnorm_residual_pair_slabs(slabs, hi, lo, p["n.norm.weight"].to(dev), p["n.norm.bias"].to(dev), mod, rps, d)
    ref = x_in.view(B, rps, d) + modulated_norm(y64.float().view(B, rps, d), lat, p, "n.")
    got = ops.pair_value(hi, lo, d).cpu()
    e = rel_l2(got, ref.view(M, d))
    print(f"pair ModulatedNorm from split-K slabs ({lo_bits}-bit low part): rel-L2 vs oracle {e:.2e}")
    assert e < 1e-5
    hi2, lo2 = ops.split_pair(x.to(dev), ld, lo_bits)
    ops.modnorm_residual_pair(ops.gemm(ad, wd), hi2, lo2, p["n.norm.weight"].to(dev), p["n.norm.bias"].to(dev), mod, rps, d)
    e_one = rel_l2(ops.pair_value(hi2, lo2, d).cpu(), got)
    assert e_one < 3e-3  # (y rounded to bf16 on that path)
    # bf16 slabs (the forecast path's default at one unit per step): every slab is the bf16 rounding of the fp32 slab, and the result
    # is as far from the exact one as the one-launch path's (each rounds the branch output at bf16 precision)
    sb = torch.full((2, M, d), float("nan"), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().swiftk_gemm_splitk_bf16(ad.data_ptr(), K, wd.data_ptr(), K, sb.data_ptr(), d, M * d, M, d, K, 2,
                                                  torch.cuda.current_stream().cuda_stream), "swiftk_gemm_splitk_bf16")
    assert torch.equal(sb, slabs.to(torch.bfloat16))
    hi3, lo3 = ops.split_pair(x.to(dev), ld, lo_bits)
    ops.modnorm_residual_pair_slabs(sb, hi3, lo3, p["n.norm.weight"].to(dev), p["n.norm.bias"].to(dev), mod, rps, d)
    e_bf = rel_l2(ops.pair_value(hi3, lo3, d).cpu(), got)
    print(f"  bf16 slabs vs fp32 slabs {e_bf:.2e} (one launch, bf16 y: {e_one:.2e})")
    assert e_bf < 3e-3 and e_bf < 2.0 * e_one


@pytest.mark.parametrize("units,K,ldk", [(3, 1056, 1088), (4, 2816, 2816), (6, 1056, 1088), (4, 1056, 1088)])
def test_gemm_tail_split_and_halves_norm(dev, units, K, ldk):
    """Round 6, small batches: wo / w2 with the persistent walk's LAST round as two k-halves (swiftk_gemm_tail_split_bf16) and the
    packed pair norm that adds the halves (swiftk_modnorm_residual_pair_halves_bf16).  Whole tiles are bit-equal to the plain GEMM and
    leave slab 1 alone, split tiles sum to the product; the norm reads slab 1 under the split tiles only and equals the one-y kernel
    on bf16(slab 0 + slab 1) bit for bit; shapes without such a round are refused."""
    import ctypes
    from swift_amd import _lib, ops
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    rps, d = 8192, 1056
    M = units * rps
    g = torch.Generator(device=dev).manual_seed(units * 1000 + K)
    a = torch.zeros(M, ldk, dtype=torch.bfloat16, device=dev)
    a[:, :K] = torch.randn(M, K, generator=g, device=dev).bfloat16()
    w = torch.zeros(d, ldk, dtype=torch.bfloat16, device=dev)
    w[:, :K] = (0.03 * torch.randn(d, K, generator=g, device=dev)).bfloat16()
    slabs = torch.full((2, M, d), float("nan"), dtype=torch.bfloat16, device=dev)
    tail = (ctypes.c_int64 * 3)(-1, -1, -1)
    _lib.check(L.swiftk_gemm_tail_split_bf16(a.data_ptr(), ldk, w.data_ptr(), ldk, slabs.data_ptr(), d, M * d, M, d, K, tail, st),
               "swiftk_gemm_tail_split_bf16")
    rows_from, tail_from, gm = tail[0], tail[1], tail[2]
    ntm, ntn = M // 256, 3
    tiles = ntm * ntn
    assert tail_from == tiles - tiles % 256 and gm >= 1 and rows_from == (tail_from // (gm * ntn)) * gm * 256 and 0 < rows_from < M
    # the walk's tile order (groups of gm tile rows, column-major inside a group): which (tile row, tile column) cells are split
    tm, tn = torch.meshgrid(torch.arange(ntm), torch.arange(ntn), indexing="ij")
    grp = tm // gm
    idx = grp * gm * ntn + tn * torch.clamp(ntm - grp * gm, max=gm) + (tm - grp * gm)
    split = (idx >= tail_from).to(dev)
    assert int(split.sum()) == tiles % 256 and not bool(split[: rows_from // 256].any())
    plain = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
    _lib.check(L.swiftk_gemm(a.data_ptr(), ldk, w.data_ptr(), ldk, plain.data_ptr(), d, M, d, K, _lib.BF16, _lib.BF16, _lib.EPI_NONE, None, None, 0, st),
               "swiftk_gemm")
    torch.cuda.synchronize()
    sel = split[:, None, :, None].expand(ntm, 256, ntn, 352)
    s0, s1, pv = slabs[0].view(ntm, 256, ntn, 352), slabs[1].view(ntm, 256, ntn, 352), plain.view(ntm, 256, ntn, 352)
    assert torch.equal(s0[~sel], pv[~sel])                     # whole tiles: the same kernel walk, the same sums ...
    assert bool(torch.isnan(s1[~sel].float()).all())           # ... and slab 1 untouched under them
    halves = s0[sel].float() + s1[sel].float()
    assert not torch.isnan(halves).any()
    e = rel_l2(halves.cpu(), pv[sel].float().cpu())
    print(f"tail split, {units} units, K {K}: {tiles % 256} of {tiles} tiles as halves (rows from {rows_from}); halves' sum vs whole-K tile {e:.2e}")
    assert e < 4e-3
    # the norm on the poisoned slabs (NaN wherever slab 1 does not exist) == the one-y kernel on bf16(slab 0 [+ slab 1])
    x = torch.randn(M, d, generator=g, device=dev)
    gam, bet = (1 + 0.1 * torch.randn(d, generator=g, device=dev)), 0.1 * torch.randn(d, generator=g, device=dev)
    mod = 0.1 * torch.randn(units, 2 * d, generator=g, device=dev)
    ld = ops.k_pad(torch.bfloat16, d)
    hi, lo = ops.split_pair(x, ld, 8)
    hi2, lo2 = hi.clone(), lo.clone()
    _lib.check(L.swiftk_modnorm_residual_pair_halves_bf16(slabs.data_ptr(), M * d, tail, hi.data_ptr(), ld, lo.data_ptr(), gam.data_ptr(),
                                                          bet.data_ptr(), mod.data_ptr(), 2 * d, M, d, rps, 1e-6, st), "halves norm")
    ysum = torch.where(sel, (s0.float() + s1.float()).bfloat16(), s0).reshape(M, d).contiguous()
    ops.modnorm_residual_pair(ysum, hi2, lo2, gam, bet, mod, rps, d)
    torch.cuda.synchronize()
    assert not torch.isnan(ops.pair_value(hi, lo, d)).any()
    assert torch.equal(hi, hi2) and torch.equal(lo, lo2)
    # tail = NULL: two halves everywhere (behind swiftk_gemm_splitk_bf16)
    both = torch.randn(2, M, d, generator=g, device=dev).bfloat16()
    hi3, lo3 = ops.split_pair(x, ld, 8)
    hi4, lo4 = hi3.clone(), lo3.clone()
    _lib.check(L.swiftk_modnorm_residual_pair_halves_bf16(both.data_ptr(), M * d, None, hi3.data_ptr(), ld, lo3.data_ptr(), gam.data_ptr(),
                                                          bet.data_ptr(), mod.data_ptr(), 2 * d, M, d, rps, 1e-6, st), "halves norm, all rows")
    ops.modnorm_residual_pair((both[0].float() + both[1].float()).bfloat16(), hi4, lo4, gam, bet, mod, rps, d)
    assert torch.equal(hi3, hi4) and torch.equal(lo3, lo4)
    # no such round: two units (192 tiles: one partly filled round), five units (480: a last round of 224 > 128), eight (768 = 3 x 256)
    for u in (2, 5, 8):
        Mu = u * rps
        au, su = torch.zeros(Mu, ldk, dtype=torch.bfloat16, device=dev), torch.empty(2, Mu, d, dtype=torch.bfloat16, device=dev)
        assert L.swiftk_gemm_tail_split_bf16(au.data_ptr(), ldk, w.data_ptr(), ldk, su.data_ptr(), d, Mu * d, Mu, d, K, tail, st) == -2  # SWIFTK_ESHAPE


@pytest.mark.parametrize("rows,cols,inter", [(3168, 1056, 0), (5632, 1056, 2816), (1056, 2816, 0), (276, 1056, 0), (70, 130, 35)])
def test_cast_pad_t_both_operands(dev, rows, cols, inter):
    """swiftk_cast_pad_t: a weight's forward operand (bf16, zero-padded rows; w1: (gate, up)-interleaved as SWIFTK_EPI_SWIGLU reads
    it, swinv2.py:96-101) and its transpose (the data-gradient GEMM's operand) in one pass -- bit-equal to torch's casts."""
    from swift_amd import ops
    w = rnd((rows, cols), 71, 0.03).to(dev)
    ko, kt = ops.k_pad(torch.bfloat16, cols), ops.k_pad(torch.bfloat16, rows)
    out = torch.full((rows, ko), 7.0, dtype=torch.bfloat16, device=dev)
    out_t = torch.full((cols, kt), 7.0, dtype=torch.bfloat16, device=dev)
    ops.cast_pad_t(w, out, out_t, inter)
    ref = w.view(2, inter, cols).permute(1, 0, 2).reshape(rows, cols) if inter else w
    assert torch.equal(out[:, :cols], ref.bfloat16()) and (out[:, cols:].float() == 0).all()
    assert torch.equal(out_t[:, :rows], ref.bfloat16().t()) and (out_t[:, rows:].float() == 0).all()


@pytest.mark.parametrize("d", [1056, 960])
@pytest.mark.parametrize("K,ldk", [(1056, 1088), (2816, 2816), (960, 960), (64, 64)])
@pytest.mark.parametrize("rows", [32, 64])
def test_gemm_modnorm_residual_pair_complete_rows(dev, rows, K, ldk, d):
    """Round 5, small batches: wo / w2 and their ModulatedNorm as ONE kernel over complete rows
    (swiftk_gemm_modnorm_residual_pair, gemm_rownorm.hip).  Its y is the bf16 rounding of the fp32-accumulated product -- the
    k order of the tiled GEMM, so bit-equal to swiftk_gemm's bf16 output -- and its update is modnorm_residual_pair's arithmetic:
    against the two-kernel path (same y; the row sums associate differently) and against the oracle's ModulatedNorm on the
    fp64 product rounded to bf16 (swinv2.py:83-86, 211-212).  K = 1056 in rows of 1088 ends half-way into a 64-wide k-tile (33
    k-steps: a partial activation super-stage); K = 64 is the two-step minimum."""
    from oracle.swinv2 import modulated_norm
    from swift_amd import ops
    B, rps = 3, 128
    M = B * rps
    a, w, x = rnd((M, ldk), 61), rnd((d, ldk), 62, 0.03), rnd((M, d), 63)
    a[:, K:] = float("nan")  # columns beyond K are never read into a product
    w[:, K:] = float("nan")
    ad, wd = to_dt(a, torch.bfloat16, dev), to_dt(w, torch.bfloat16, dev)
    p = {"n.norm.weight": 1 + 0.1 * rnd((d,), 14), "n.norm.bias": 0.1 * rnd((d,), 15),
         "n.modulation.weight": 0.02 * rnd((2 * d, d), 16), "n.modulation.bias": 0.1 * rnd((2 * d,), 17)}
    lat = rnd((B, d), 18)
    mod = torch.nn.functional.linear(lat, p["n.modulation.weight"], p["n.modulation.bias"]).to(dev)
    g, b = p["n.norm.weight"].to(dev), p["n.norm.bias"].to(dev)
    ld = ops.k_pad(torch.bfloat16, d)
    hi, lo = ops.split_pair(x.to(dev), ld, 8)
    hi[:, d:] = 7.0
    x_in = ops.pair_value(hi, lo, d).cpu()
    ops.gemm_modnorm_residual_pair(ad, wd, hi, lo, g, b, mod, rps, d, k=K, rows_per_workgroup=rows)
    got = ops.pair_value(hi, lo, d).cpu()
    assert torch.isfinite(got).all() and (hi[:, d:].float() == 7.0).all()
    # the two-kernel path on the same operands
    hi2, lo2 = ops.split_pair(x.to(dev), ld, 8)
    a0, w0 = ad.clone(), wd.clone()
    a0[:, K:] = 0
    w0[:, K:] = 0
    y = ops.gemm(a0, w0)  # (zero k-padding adds nothing: the same fp32 sums)
    ops.modnorm_residual_pair(y, hi2, lo2, g, b, mod, rps, d)
    e2 = rel_l2(got, ops.pair_value(hi2, lo2, d).cpu())
    diff_hi = float((hi2[:, :d] != hi[:, :d]).float().mean())
    y64 = ad[:, :K].float().cpu().double() @ wd[:, :K].float().cpu().double().T
    ref = x_in.view(B, rps, d) + modulated_norm(y64.float().bfloat16().float().view(B, rps, d), lat, p, "n.")
    e = rel_l2(got, ref.view(M, d))
    print(f"complete-row GEMM + norm ({rows} rows, K {K}, d {d}): vs GEMM + norm kernels {e2:.2e} (hi differs on {diff_hi:.1e}), vs oracle {e:.2e}")
    assert e2 < 2e-6 and diff_hi < 1e-3
    assert e < 2e-4  # (the fp64 product rounds to another bf16 than the fp32-accumulated one on ~1e-3 of the elements)


def test_gemm_modnorm_residual_pair_rejects(dev):
    from swift_amd import _lib, ops
    d, K, M = 1056, 1056, 128
    a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(d, K, dtype=torch.bfloat16, device=dev)
    hi = torch.zeros(M, 1088, dtype=torch.bfloat16, device=dev)
    lo = torch.zeros(M, d, dtype=torch.uint8, device=dev)
    g = torch.ones(d, device=dev)
    mod = torch.zeros(2, 2 * d, device=dev)
    for kw, exc in (({"rows_per_workgroup": 48}, "SHAPE"), ({"k": 1040}, "SHAPE"), ({"k": 32}, "SHAPE")):
        with pytest.raises(RuntimeError):
            ops.gemm_modnorm_residual_pair(a, w, hi, lo, g, g, mod, 64, d, **kw)
    with pytest.raises(RuntimeError):  # rows of a workgroup must be of one sample
        ops.gemm_modnorm_residual_pair(a, w, hi, lo, g, g, mod, 48, d, rows_per_workgroup=32)
    with pytest.raises(RuntimeError):  # another width
        ops.gemm_modnorm_residual_pair(a, torch.zeros(1280, K, dtype=torch.bfloat16, device=dev), torch.zeros(M, 1280, dtype=torch.bfloat16, device=dev),
                                       torch.zeros(M, 1280, dtype=torch.uint8, device=dev), torch.ones(1280, device=dev),
                                       torch.ones(1280, device=dev), torch.zeros(2, 2560, device=dev), 64, 1280)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_patchify_three_sources(dev, dt):
    from oracle.swinv2 import patchify
    from swift_amd import ops
    B, H, W = 2, 32, 64
    a, b, c = rnd((B, 5, H, W), 17), rnd((B, 4, H, W), 18), rnd((B, 3, H, W), 19)
    lda = ops.k_pad(dt, 12 * 4)
    out = ops.patchify([a.to(dev), b.to(dev), c.to(dev)], [0.5, 1.0, 2.0], (2, 2), lda, dt)
    ref = patchify(torch.cat([a * 0.5, b, c * 2.0], 1), (2, 2)).reshape(-1, 48)
    assert rel_l2(out[:, :48].float().cpu(), ref) < (1e-7 if dt == torch.float32 else 3e-3)
    assert (out[:, 48:].float() == 0).all()


def test_unpatchify_affine(dev):
    from oracle.swinv2 import unpatchify
    from swift_amd import ops
    B, C, H, W = 2, 5, 32, 64
    tok = rnd((B, 16 * 32, 20), 20)
    xt, al, be = rnd((B, C, H, W), 21), torch.tensor([0.3, -1.0]), torch.tensor([2.0, 0.5])
    out = ops.unpatchify_affine(tok.to(dev), (B, C, H, W), (2, 2), xt.to(dev), al.to(dev), be.to(dev))
    ref = al.view(B, 1, 1, 1) * xt + be.view(B, 1, 1, 1) * unpatchify(tok, (2, 2), (16, 32))
    assert rel_l2(out.cpu(), ref) < 1e-7
    out = ops.unpatchify_affine(tok.to(dev), (B, C, H, W), (2, 2))
    assert torch.equal(out.cpu(), unpatchify(tok, (2, 2), (16, 32)))


def test_timestep_embed_and_small_linear(dev):
    from oracle.swinv2 import timestep_embedding
    from swift_amd import ops
    d, B = 1056, 5
    t = torch.tensor([0.0, 1.0, math.pi / 2, 0.37, 1.5])
    aux = torch.tensor([[0.6], [1.2], [2.4], [0.6], [0.0]])
    aw, ab = rnd((d, 1), 22, 0.02), rnd((d,), 23, 0.02)
    half = d // 2
    freqs = torch.exp(-math.log(10_000) * torch.arange(half, dtype=torch.float32) / half)
    e = ops.timestep_embed(t.to(dev), aux.to(dev), freqs.to(dev), aw.to(dev), ab.to(dev), d, 1.0)
    ref = timestep_embedding(t, d) + torch.nn.functional.linear(aux, aw, ab)
    assert rel_l2(e.cpu(), ref) < 2e-6
    w, b = rnd((2112, d), 24, 0.03), rnd((2112,), 25)
    x = rnd((B, d), 26)
    o = ops.linear_small(x.to(dev), w.to(dev), b.to(dev), act=1)
    assert rel_l2(o.cpu(), torch.nn.functional.silu(x.double() @ w.double().T + b.double())) < 2e-6
    o = ops.linear_small(x.to(dev), w.to(dev), None, act=0)
    assert rel_l2(o.cpu(), x.double() @ w.double().T) < 2e-6
    # wide outputs (the 24 modulation Linears as one matrix): x staged in LDS, the next weight row prefetched (K <= 1280) or
    # the general walk (K = 1408); one unit, a ragged batch, and more than one LDS batch of 8
    for K, Bn in ((1056, 1), (1056, 5), (1056, 11), (1408, 3)):
        w, b, x = rnd((8448 + 4, K), 32, 0.03), rnd((8448 + 4,), 33), rnd((Bn, K), 34)
        o = ops.linear_small(x.to(dev), w.to(dev), b.to(dev), act=0)
        assert rel_l2(o.cpu(), x.double() @ w.double().T + b.double()) < 2e-6, (K, Bn)


def test_rollout_update_and_axpby(dev):
    from swift_amd import ops
    B, C, H, W = 2, 7, 16, 32
    x, y = rnd((B, C, H, W), 27), rnd((B, C, H, W), 28)
    mx, sx, st = rnd((C,), 29, 3.0), rnd((C,), 30).abs() + 0.5, rnd((C,), 31).abs() + 0.1
    xd, phys = x.to(dev).clone(), torch.empty(B, C, H, W, device=dev)
    ops.rollout_update(xd, y.to(dev), mx.to(dev), sx.to(dev), st.to(dev), phys=phys)
    v = lambda s: s.view(1, C, 1, 1)
    p = (x * v(sx) + v(mx)) + y * v(st)
    assert rel_l2(phys.cpu(), p) < 1e-7
    assert rel_l2(xd.cpu(), (p - v(mx)) / v(sx)) < 1e-6
    o = ops.axpby(0.25, x.to(dev), -1.5, y.to(dev))
    assert rel_l2(o.cpu(), 0.25 * x - 1.5 * y) < 1e-7
    # a channel the dataset forces to zero in standardised form (zero_field: sea_surface_temperature) travels as std_x = 0 --
    # with --interval 24 its residual still reaches the physical output (data/era5.py:135-170, generate.py:120-131)
    mx2, sx2 = mx.clone(), sx.clone()
    mx2[3], sx2[3] = 0.0, 0.0
    xd, phys = x.to(dev).clone(), torch.empty(B, C, H, W, device=dev)
    ops.rollout_update(xd, y.to(dev), mx2.to(dev), sx2.to(dev), st.to(dev), phys=phys)
    assert torch.equal(phys[:, 3].cpu(), y[:, 3] * st[3]) and float(xd[:, 3].abs().sum()) == 0.0
    keep = [c for c in range(C) if c != 3]
    assert rel_l2(phys[:, keep].cpu(), p[:, keep]) < 1e-7 and torch.isfinite(xd).all()


def test_rejects_cpu_tensors_and_bad_shapes(dev):
    from swift_amd import ops
    from swift_amd._lib import SwiftkError
    with pytest.raises(SwiftkError):
        ops.gemm(torch.zeros(4, 64), torch.zeros(4, 64))
    with pytest.raises(SwiftkError):  # K not a multiple of the k-tile
        ops.gemm(torch.zeros(4, 40, device=dev), torch.zeros(4, 40, device=dev))
    with pytest.raises(SwiftkError):  # grid not divisible by the 16x16 window
        ops.window_attention(torch.zeros(1, 24 * 16, 3168, device=dev), torch.zeros(12, device=dev), (24, 16), 12)


def _prenorm_reference(qkv, scale, heads, hd):
    """(q-hat | k-hat | v) laid out like qkv, in fp32: what SWIFTK_EPI_QKNORM must produce (swinv2.py:123-127)."""
    B, n, _ = qkv.shape
    v = qkv.reshape(B, n, heads, 3, hd).clone()
    tau = torch.clamp(scale, max=math.log(100.0)).exp().view(1, 1, heads, 1)
    v[:, :, :, 0] = v[:, :, :, 0] / v[:, :, :, 0].norm(dim=-1, keepdim=True).clamp_min(1e-12) * tau
    v[:, :, :, 1] = v[:, :, :, 1] / v[:, :, :, 1].norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return v.reshape(B, n, -1)


@pytest.mark.parametrize("dt,hd", [(torch.float32, 88), (torch.bfloat16, 88), (torch.bfloat16, 80), (torch.bfloat16, 96),
                                   (torch.float32, 80), (torch.float32, 96)])
def test_gemm_qknorm_epilogue(dev, dt, hd):
    """head_dim 80 / 96 (dim 1280 / 1536 with 16 heads in the reference's larger variants): 320- / 384-wide GEMM tiles, with bf16
    and (round 5: the exact engine and the split engine's hot head pairs on those variants) fp32 operands."""
    from swift_amd import ops
    M, heads = 1024, 12
    K = ops.k_pad(dt, heads * hd)
    a, w = rnd((M, K), 40), rnd((3 * heads * hd, K), 41, 0.03)
    scale = torch.log(torch.tensor([10.0, 3.0, 30.0, 200.0, 1.0, 10.0, 50.0, 99.0, 101.0, 5.0, 20.0, 10.0]))
    ad, wd = to_dt(a, dt, dev), to_dt(w, dt, dev)
    c = ops.gemm(ad, wd, epilogue=ops.EPI_QKNORM, bias=scale.to(dev), head_dim=hd)
    raw = (ad.float().cpu().double() @ wd.float().cpu().double().T).float()
    ref = _prenorm_reference(raw.view(1, M, -1), scale, heads, hd)[0]
    assert rel_l2(c.float().cpu(), ref) < (F32_TOL if dt == torch.float32 else 4e-3)


@pytest.mark.parametrize("dt,flags,hd", [(torch.bfloat16, 1, 88), (torch.bfloat16, 3, 88), (torch.float32, 1, 88),
                                         (torch.bfloat16, 1, 80), (torch.bfloat16, 1, 96)])
@pytest.mark.parametrize("shift", [(0, 0), (8, 8)])
@pytest.mark.parametrize("B", [1, 3, 8])  # 8: 576 items > 256 workgroups, the steady state of the streamed kernel
def test_window_attention_prenormalised(dev, dt, flags, shift, B, hd):
    """flags 1 = PRENORM (bf16: persistent LDS-DMA-pipelined kernel, head_dim 80 / 88 / 96), 3 = PRENORM|NO_PIPE (per-item)."""
    from oracle.swinv2 import window_token_index
    from swift_amd import ops
    grid, heads = (32, 48), 12
    n = grid[0] * grid[1]
    qkv = rnd((B, n, 3 * heads * hd), 42 + B)
    scale = torch.log(torch.tensor([10.0, 3.0, 30.0, 200.0, 1.0, 10.0, 50.0, 99.0, 101.0, 5.0, 20.0, 10.0]))
    pre = to_dt(_prenorm_reference(qkv, scale, heads, hd), dt, dev)
    out = ops.window_attention(pre, None, grid, heads, shift, flags=flags)
    idx = window_token_index(grid, (16, 16), shift)
    src = pre.float().cpu()[:, idx.reshape(-1)].reshape(B * idx.shape[0], 256, heads, 3, hd).permute(0, 2, 1, 3, 4)
    ow = (src[..., 0, :] @ src[..., 1, :].transpose(-2, -1)).softmax(-1) @ src[..., 2, :]  # b h n d
    ref = torch.empty(B, n, heads * hd)
    ref[:, idx.reshape(-1)] = ow.permute(0, 2, 1, 3).reshape(B, n, -1)
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out.float().cpu(), ref) < (2e-5 if dt == torch.float32 else 1.2e-2)


@pytest.mark.parametrize("hd", [88, 80, 96])
@pytest.mark.parametrize("shift", [(0, 0), (8, 8), (3, 5)])
@pytest.mark.parametrize("B", [1, 3, 8])  # 8: several tiles per GEMM workgroup and several items per attention workgroup
def test_window_tiled_qkv_path(dev, shift, B, hd):
    """to_qkv stored window-tiled (swiftk_gemm_qkv_tiled) + SWIFTK_ATTN_TILED attention: the tiled tensor is an exact
    permutation of the row-major QK-norm GEMM output (window_partition of the rolled grid, swinv2.py:17-26,185-189),
    attention over it is bit-identical to attention over the row-major tensor, and both match the fp32 formula.
    The scale vector mixes heads whose logit bound is <= 48 (max-free streaming softmax) and > 48 (online form)."""
    from oracle.swinv2 import window_token_index
    from swift_amd import ops
    grid, heads = (32, 48), 12
    n, d = grid[0] * grid[1], heads * hd
    K = ops.k_pad(torch.bfloat16, d)
    a, w = rnd((B * n, K), 50 + B), rnd((3 * heads * hd, K), 51, 0.03)
    a[:, d:] = 0
    scale = torch.log(torch.tensor([10.0, 3.0, 30.0, 200.0, 1.0, 10.0, 50.0, 99.0, 101.0, 5.0, 20.0, 48.0])).to(dev)
    ad, wd = to_dt(a, torch.bfloat16, dev), to_dt(w, torch.bfloat16, dev)
    c = ops.gemm(ad, wd, epilogue=ops.EPI_QKNORM, bias=scale, head_dim=hd)
    ct = ops.gemm_qkv_tiled(ad, wd, scale, B, grid, heads, shift, head_dim=hd)
    idx = window_token_index(grid, (16, 16), shift)  # [windows, 256] token of each window slot
    perm = c.view(B, n, heads, 3, hd)[:, idx.reshape(-1).to(dev)].view(B, idx.shape[0], 256, heads, 3, hd)
    assert torch.equal(ct, perm.permute(0, 1, 3, 4, 2, 5).contiguous())
    # K ending half-way into the last k-tile (1056 of 1088) gives the same numbers
    assert torch.equal(ct, ops.gemm_qkv_tiled(ad, wd, scale, B, grid, heads, shift, k=d, head_dim=hd))
    out_rm = ops.window_attention(c.view(B, n, -1), scale, grid, heads, shift, flags=ops.ATTN_PRENORM)
    out_t = ops.window_attention_tiled(ct, scale, grid, heads, shift)
    assert torch.equal(out_rm, out_t)
    src = c.float().cpu().view(B, n, -1)[:, idx.reshape(-1)].reshape(B * idx.shape[0], 256, heads, 3, hd).permute(0, 2, 1, 3, 4)
    ow = (src[..., 0, :] @ src[..., 1, :].transpose(-2, -1)).softmax(-1) @ src[..., 2, :]
    ref = torch.empty(B, n, heads * hd)
    ref[:, idx.reshape(-1)] = ow.permute(0, 2, 1, 3).reshape(B, n, -1)
    assert torch.isfinite(out_t.float()).all()
    assert rel_l2(out_t.float().cpu(), ref) < 1.2e-2


@pytest.mark.parametrize("hd,shift,B", [(88, s_, b_) for s_ in ((0, 0), (8, 8), (3, 5)) for b_ in (1, 3, 8)] +
                         # head_dim 80 / 96: the reference's 468 M / 664 M variants (16 heads; era5-swinv2-1.4-scm.yaml:29-36) --
                         # 20 / 24 k-tiles (an even count), no half k-tile, head_dim 96 with the LDS overlap and the VALU row sum
                         [(h_, s_, b_) for h_ in (80, 96) for s_ in ((0, 0), (3, 5)) for b_ in (1, 8)])
def test_fused_qkv_attention(dev, hd, shift, B):  # B = 8: several items per workgroup (cross-item prefetch, output hand-over)
    """swiftk_qkv_attention_fused (to_qkv + cosine norm + shifted-window attention in one kernel, q/k/v never in HBM)
    against (a) the two-kernel path it replaces -- same bf16 operands, same fp32 accumulation, so the outputs agree to the
    rounding of the normalised q/k/v to bf16 and of the probabilities -- and (b) the fp32 formula on the QK-norm GEMM's
    output (swinv2.py:119-136).  Heads with logit bound <= 48 (max-free softmax) and > 48 (online form) are mixed."""
    from oracle.swinv2 import window_token_index
    from swift_amd import ops
    grid, heads = (32, 48), 12 if hd == 88 else 16
    n, d = grid[0] * grid[1], heads * hd
    K = ops.k_pad(torch.bfloat16, d)
    a, w = rnd((B * n, K), 60 + B), rnd((3 * heads * hd, K), 61, 0.03)
    a[:, d:] = 0
    w[:, d:] = 0
    scale = torch.log(torch.tensor([10.0, 3.0, 30.0, 200.0, 1.0, 10.0, 50.0, 99.0, 101.0, 5.0, 20.0, 48.0, 2.0, 60.0, 47.0, 49.0][:heads])).to(dev)
    ad, wd = to_dt(a, torch.bfloat16, dev), to_dt(w, torch.bfloat16, dev)
    ct = ops.gemm_qkv_tiled(ad, wd, scale, B, grid, heads, shift, k=d, head_dim=hd)
    two = ops.window_attention_tiled(ct, scale, grid, heads, shift)
    ldo = ops.k_pad(torch.bfloat16, d) + (0 if hd == 88 else 64)  # padded rows, as in the engine (88: 1056 -> 1088)
    out = torch.full((B, n, ldo), 7.0, dtype=torch.bfloat16, device=dev)
    ops.qkv_attention_fused(ad, wd, scale, B, grid, heads, shift, out=out[..., :d], k=d, head_dim=hd)
    assert torch.isfinite(out.float()).all() and (out[..., d:].float() == 7.0).all()  # pad columns untouched
    fused = out[..., :d]
    assert rel_l2(fused.float().cpu(), two.float().cpu()) < 6e-3
    # second call into the same buffer: bit-identical (no dependence on what the LDS / the output held before)
    out2 = torch.zeros_like(out)
    ops.qkv_attention_fused(ad, wd, scale, B, grid, heads, shift, out=out2[..., :d], k=d, head_dim=hd)
    assert torch.equal(out2[..., :d], fused)
    c = ops.gemm(ad[:, :d], wd[:, :d], epilogue=ops.EPI_QKNORM, bias=scale, head_dim=hd)
    idx = window_token_index(grid, (16, 16), shift)
    src = c.float().cpu().view(B, n, -1)[:, idx.reshape(-1)].reshape(B * idx.shape[0], 256, heads, 3, hd).permute(0, 2, 1, 3, 4)
    ow = (src[..., 0, :] @ src[..., 1, :].transpose(-2, -1)).softmax(-1) @ src[..., 2, :]
    ref = torch.empty(B, n, heads * hd)
    ref[:, idx.reshape(-1)] = ow.permute(0, 2, 1, 3).reshape(B, n, -1)
    assert rel_l2(fused.float().cpu(), ref) < 1.2e-2
    # (c) the ORACLE as the yardstick, nothing from the HIP library in the reference value: to_qkv by F.linear on the same
    # bf16-valued operands, then oracle.cosine_window_attention over the oracle's window index map (swinv2.py:119-136,
    # 185-208) -- in exact fp32 (distance = the kernel's bf16 rounding of q-hat / k-hat / v / P) and with the oracle's
    # bf16-operand emulation (distance = two independent roundings of the same quantities)
    import torch.nn.functional as F
    from oracle.swinv2 import cosine_window_attention
    af, wf = ad[:, :d].float().cpu(), wd[:, :d].float().cpu()
    qkv = F.linear(af, wf).view(B, n, -1)
    sc = scale.cpu().view(1, heads, 1, 1)
    for emu, tol in ((False, 1.2e-2), (True, 1.2e-2)):
        ow = cosine_window_attention(qkv[:, idx.reshape(-1)].reshape(B * idx.shape[0], 256, -1), sc, heads, naive=True,
                                     emulate_bf16=emu)
        oref = torch.empty(B, n, heads * hd)
        oref[:, idx.reshape(-1)] = ow.reshape(B, idx.numel(), -1)
        e = rel_l2(fused.float().cpu(), oref)
        print(f"fused to_qkv + attention vs oracle (emulate_bf16={emu}), head_dim {hd}, shift {shift}, B {B}: rel-L2 {e:.3e}")
        assert e < tol


def test_fused_qkv_attention_rejects_what_it_cannot_run(dev):
    """The fused kernel is built for head_dim 80 / 88 / 96 and 16x16 windows; anything else is refused with an error code (the
    forward then takes the two-kernel path) -- never a silent wrong answer."""
    from swift_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    a = torch.zeros(2 * 512, 1152, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(3168, 1152, dtype=torch.bfloat16, device=dev)
    sc = torch.zeros(12, device=dev)
    out = torch.zeros(2 * 512, 1088, dtype=torch.bfloat16, device=dev)
    call = lambda K, hd, gh, gw, sh=0, sw=0, ldo=1088: L.swiftk_qkv_attention_fused(
        a.data_ptr(), 1152, w.data_ptr(), 1152, sc.data_ptr(), out.data_ptr(), ldo, K, 2, gh, gw, 12, hd, sh, sw, st)
    assert call(1056, 88, 16, 32) == 0                      # 16.5 k-tiles: runs
    assert call(1152, 88, 16, 32) == 0                      # 18 k-tiles (an even count): runs
    assert call(1056, 64, 16, 32) == -2                     # head_dim
    assert call(64, 88, 16, 32) == -2                       # a single k-tile
    assert call(1056, 88, 24, 32) == -2                     # grid not a multiple of the window
    assert call(1056, 88, 16, 32, sh=16) == -2              # shift outside the grid
    assert call(1056, 88, 16, 32, ldo=1000) == -2           # output rows too short for 12 x 88 columns
    assert L.swiftk_qkv_attention_fused(None, 1152, w.data_ptr(), 1152, sc.data_ptr(), out.data_ptr(), 1088, 1056, 2, 16, 32, 12,
                                        88, 0, 0, st) == -1
    torch.cuda.synchronize()


def test_ensemble_metrics_vs_reference_golden(dev):
    """swiftk_ensemble_sums -> RMSE / CRPS / spread-skill against the reference's eval/metrics.py functions (golden)."""
    from swift_amd.eval.metrics import all_metrics, lat_weighted_crps
    g = load_golden("metrics_tiny")
    pred, y = torch.from_numpy(g["pred"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    names = [f"v{i}" for i in range(4)]
    got = all_metrics(pred, y, names, g["lat"], "6h")
    ref = dict(zip([str(k) for k in g["keys"]], g["values"]))
    assert set(got) == set(ref)
    for k, v in ref.items():
        assert float(got[k]) == pytest.approx(v, rel=5e-5), k
    assert set(lat_weighted_crps(pred, y, names, g["lat"], "6h")) == {f"crps_v{i}_6h" for i in range(4)}
    # 12 members (the rollout's ensemble size) against the fp64 formulas
    from oracle import metrics as omet
    p12, y12 = rnd((2, 12, 3, 32, 64), 70).to(dev), rnd((2, 3, 32, 64), 71).to(dev)
    lat = np.linspace(-88, 88, 32)
    got = all_metrics(p12, y12, ["a", "b", "c"], lat, "x")
    for name, fn in (("rmse", omet.rmse), ("crps", omet.crps), ("ssr", omet.spread_skill_ratio)):
        r = fn(p12.cpu().double(), y12.cpu().double(), lat)
        for i, v in enumerate("abc"):
            assert float(got[f"{name}_{v}_x"]) == pytest.approx(float(r[i]), rel=5e-5)


@pytest.mark.parametrize("d,rps,copy", [(1056, 8192, True), (1056, 8192, False), (1280, 4096, True), (96, 64, True)])
def test_modnorm_residual_split3_is_norm_plus_split_bit_for_bit(dev, d, rps, copy):
    """Round 6, split engine: swiftk_modnorm_residual_split3 = swiftk_modnorm_residual (fp32, swinv2.py:83-86,211-212) followed by
    swiftk_split3(order 0) of the new rows, in one pass -- x, the optional fp32 copy and the [hi | lo | hi] operand blocks with their
    zero k-padding must be BIT-EQUAL to the two-step form."""
    from swift_amd import _lib, ops
    L = _lib.lib()
    B = 2
    M = B * rps
    y, x0 = rnd((M, d), 81).to(dev), rnd((M, d), 82).to(dev)
    gamma, beta = (1 + 0.1 * rnd((d,), 83)).to(dev), (0.1 * rnd((d,), 84)).to(dev)
    mod = (0.3 * rnd((B, 2 * d), 85)).to(dev)
    kd, ld3 = ops.k_pad(torch.float32, d), ops.k_pad(torch.bfloat16, 3 * d)
    st = torch.cuda.current_stream().cuda_stream
    xa, ca = x0.clone(), torch.full((M, kd), 7.0, device=dev)
    _lib.check(L.swiftk_modnorm_residual(y.data_ptr(), d, xa.data_ptr(), ca.data_ptr(), kd, gamma.data_ptr(), beta.data_ptr(), mod.data_ptr(), 2 * d,
                                         M, d, rps, 1e-6, _lib.F32, st), "modnorm")
    a3 = torch.full((M, ld3), 3.0, dtype=torch.bfloat16, device=dev)
    _lib.check(L.swiftk_split3(ca.data_ptr(), kd, a3.data_ptr(), ld3, M, d, 0, st), "split3")
    xb, cb = x0.clone(), torch.full((M, kd), 7.0, device=dev)
    b3 = torch.full((M, ld3), 3.0, dtype=torch.bfloat16, device=dev)
    _lib.check(L.swiftk_modnorm_residual_split3(y.data_ptr(), d, xb.data_ptr(), cb.data_ptr() if copy else None, kd, b3.data_ptr(), ld3, gamma.data_ptr(),
                                                beta.data_ptr(), mod.data_ptr(), 2 * d, M, d, rps, 1e-6, st), "modnorm_split3")
    assert torch.equal(xa, xb) and torch.equal(a3.view(torch.int16), b3.view(torch.int16))
    assert torch.equal(ca[:, :d], cb[:, :d]) if copy else bool((cb == 7.0).all())
    # shapes the chunk kernel does not take are refused (the forward then runs the two-step form)
    assert L.swiftk_modnorm_residual_split3(y.data_ptr(), d, xb.data_ptr(), None, kd, b3.data_ptr(), ld3, gamma.data_ptr(), beta.data_ptr(), mod.data_ptr(),
                                            2 * d, M, d, rps + 8, 1e-6, st) == -2


@pytest.mark.parametrize("hd,npairs", [(88, 1), (88, 3), (96, 2), (80, 1)])
def test_gemm_qknorm_qk_only_form_equals_the_full_form_on_q_and_k(dev, hd, npairs):
    """Round 6, split engine: SWIFTK_EPI_QKNORM with pos_rows = -head_dim recomputes only the [q | k] column pairs of head pairs whose
    weights sit as [q | k | v] row triples (to_qkv.weight as stored, swinv2.py:119-127) -- bit-equal to the full form on the q and k
    columns, the v columns of the output untouched."""
    from swift_amd import _lib
    L = _lib.lib()
    M, K = 2048, 1056 if hd == 88 else 16 * hd
    heads = 2 * npairs
    a = rnd((M, K), 95).to(dev)
    w = (0.03 * rnd((3 * heads * hd, K), 96)).to(dev)
    scale = torch.log(torch.tensor([10.0, 60.0] * npairs)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    full = torch.zeros(M, 3 * heads * hd, device=dev)
    _lib.check(L.swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, full.data_ptr(), 3 * heads * hd, M, 3 * heads * hd, K, _lib.F32, _lib.F32,
                             _lib.EPI_QKNORM, scale.data_ptr(), None, hd, st), "full")
    part = torch.full((M, 3 * heads * hd), 7.0, device=dev)
    _lib.check(L.swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, part.data_ptr(), 3 * heads * hd, M, 2 * heads * hd, K, _lib.F32, _lib.F32,
                             _lib.EPI_QKNORM, scale.data_ptr(), None, -hd, st), "qk only")
    f3, p3 = full.view(M, heads, 3, hd), part.view(M, heads, 3, hd)
    assert torch.equal(f3[:, :, :2], p3[:, :, :2]) and bool((p3[:, :, 2] == 7.0).all())
    assert float(f3[:, :, 0].norm(dim=-1).min()) > 5.0  # (q rows carry their logit scale: the epilogue did run)
    # refused where it cannot apply: bf16 operands
    assert L.swiftk_gemm(a.bfloat16().data_ptr(), K, w.bfloat16().data_ptr(), K, part.data_ptr(), 3 * heads * hd, M, 2 * heads * hd, K, _lib.BF16,
                         _lib.F32, _lib.EPI_QKNORM, scale.data_ptr(), None, -hd, st) == -2


@pytest.mark.parametrize("hd,heads", [(88, 12), (96, 4), (80, 2)])
def test_gemm_qknorm_single_head_form_equals_the_full_form(dev, hd, heads):
    """Round 6, split engine: SWIFTK_EPI_QKNORM with fp32 operands takes ONE head (N = 3 head_dim: one tile column whose fourth vector
    is empty) -- what recomputes a hot head alone.  Every head, the LAST one included (no logit scale is read past the array), must
    come out bit-equal to its columns of the whole-matrix call, the other columns untouched."""
    from swift_amd import _lib
    L = _lib.lib()
    M, K = 1024, 1056 if hd == 88 else 16 * hd
    a = rnd((M, K), 98).to(dev)
    w = (0.03 * rnd((3 * heads * hd, K), 99)).to(dev)
    scale = torch.log(torch.linspace(8.0, 90.0, heads)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    N = 3 * heads * hd
    full = torch.zeros(M, N, device=dev)
    _lib.check(L.swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, full.data_ptr(), N, M, N, K, _lib.F32, _lib.F32, _lib.EPI_QKNORM, scale.data_ptr(), None,
                             hd, st), "full")
    for h in (0, heads // 2, heads - 1):
        part = torch.full((M, N), 7.0, device=dev)
        c0 = h * 3 * hd
        _lib.check(L.swiftk_gemm(a.data_ptr(), K, w.data_ptr() + c0 * K * 4, K, part.data_ptr() + c0 * 4, N, M, 3 * hd, K, _lib.F32, _lib.F32,
                                 _lib.EPI_QKNORM, scale.data_ptr() + 4 * h, None, hd, st), "one head")
        assert torch.equal(part[:, c0:c0 + 3 * hd], full[:, c0:c0 + 3 * hd])
        rest = torch.ones(N, dtype=torch.bool)
        rest[c0:c0 + 3 * hd] = False
        assert bool((part[:, rest.to(dev)] == 7.0).all())


@pytest.mark.parametrize("hd,heads,shift", [(88, 12, (8, 8)), (80, 4, (0, 0)), (96, 2, (3, 5)), (64, 2, (0, 0))])
def test_window_attention_fp32_with_split_pv_vs_exact(dev, hd, heads, shift):
    """Round 6, split engine: SWIFTK_ATTN_PV_BF16X3 keeps q k^T and the softmax on the exact-fp32 MFMA and runs O = P V as three bf16
    products of (hi, lo)-split operands (swinv2.py:129-136).  Against the all-fp32 kernel: 2^-17-grade (the dropped lo x lo term and
    the operands' 16-bit mantissas), i.e. at the level of the split GEMMs' own 4.5e-6; one logit scale at the clamp."""
    from swift_amd import _lib, ops
    L = _lib.lib()
    B, gh, gw = 2, 32, 32
    d = heads * hd
    M = B * gh * gw
    qkv = rnd((M, 3 * d), 97).to(dev)
    scale = torch.log(torch.tensor([10.0] * (heads - 1) + [100.0])).to(dev)
    kd = ops.k_pad(torch.float32, d)
    st = torch.cuda.current_stream().cuda_stream
    ref, got = torch.zeros(M, kd, device=dev), torch.zeros(M, kd, device=dev)
    _lib.check(L.swiftk_window_attention(qkv.data_ptr(), 3 * d, ref.data_ptr(), kd, scale.data_ptr(), B, gh, gw, heads, hd, shift[0], shift[1],
                                         _lib.F32, 0, st), "attention")
    _lib.check(L.swiftk_window_attention(qkv.data_ptr(), 3 * d, got.data_ptr(), kd, scale.data_ptr(), B, gh, gw, heads, hd, shift[0], shift[1],
                                         _lib.F32, 8, st), "attention, split P V")
    err = rel_l2(got[:, :d].cpu(), ref[:, :d].cpu())
    worst = float((got - ref)[:, :d].abs().max() / ref[:, :d].abs().max())
    print(f"  fp32 attention, P V as three bf16 products vs exact (head_dim {hd}): rel-L2 {err:.2e}, max {worst:.2e}")
    assert 0 < err < 1.5e-5 and worst < 6e-5


def test_store_to_store_evaluation_cli_on_the_device(dev, tmp_path):
    """``python -m swift_amd.eval.metrics --truth T.zarr --pred P.zarr`` (reference eval/metrics.py:157-280) through the real
    ``swiftk_ensemble_sums``: a 12-member forecast store with a levelled variable against the oracle's restatement of the reference
    functions, and the structured evaluation_metrics.json beside the store."""
    import json
    from oracle import metrics as omet
    from swift_amd.eval import metrics as em
    from swift_amd.utils import zarrlite
    rng = np.random.default_rng(11)
    names = ["2m_temperature", "temperature_500", "temperature_850", "temperature_1000"]
    H, W, B, N, steps = 32, 64, 2, 12, 3
    lat = np.linspace(-88, 88, H)
    t_all = np.datetime64("2021-06-01T00") + np.arange(12) * np.timedelta64(6, "h")
    pred = str(tmp_path / "out" / "output-2i-3s-12m-6h.zarr")
    os.makedirs(os.path.dirname(pred))
    ch = zarrlite.create_forecast_store(pred, names, t_all[[1, 4]], lat, np.arange(W) * 5.625, members=N, steps=steps, interval=6)
    traj = rng.standard_normal((B, N, steps + 1, 4, H, W)).astype(np.float32)
    for b in range(B):
        for n in range(N):
            zarrlite.write_unit(pred, ch, b, n, traj[b, n])
    truth = str(tmp_path / "truth.zarr")
    zarrlite.create_group(truth)
    zarrlite.write_full(truth, "time", t_all.astype("datetime64[ns]").astype(np.int64), ["time"], {"units": "nanoseconds since 1970-01-01"})
    zarrlite.write_full(truth, "latitude", lat.astype(np.float32), ["latitude"])
    f2, f3 = rng.standard_normal((12, H, W)).astype(np.float32), rng.standard_normal((12, 3, H, W)).astype(np.float32)
    zarrlite.write_full(truth, "2m_temperature", f2, ["time", "latitude", "longitude"])
    zarrlite.write_full(truth, "temperature", f3, ["time", "level", "latitude", "longitude"])
    flat = em.main(["--truth", truth, "--pred", pred])
    doc = json.load(open(os.path.join(os.path.dirname(pred), "evaluation_metrics.json")))
    assert set(doc) == {"metadata", "metrics"} and doc["metadata"]["truth_file"] == truth and set(doc["metrics"]) == {"rmse", "crps", "ssr"}
    assert set(doc["metrics"]["crps"]) == {"0", "6", "12", "18"} and len(doc["metrics"]["rmse"]["18"]) == 4
    idx = np.array([1, 4])
    for j in range(steps + 1):
        P = torch.from_numpy(traj[:, :, j]).double()
        Y = torch.cat([torch.from_numpy(f2[idx + j])[:, None], torch.from_numpy(f3[idx + j])], 1).double()
        for name, fn in (("rmse", omet.rmse), ("crps", omet.crps), ("ssr", omet.spread_skill_ratio)):
            r = fn(P, Y, lat)
            for i, v in enumerate(["2m_temperature", "temperature_50", "temperature_100", "temperature_150"]):  # (levels named by position)
                assert flat[f"{name}_{v}_{6 * j}h"] == pytest.approx(float(r[i]), rel=5e-5), (name, v, j)
                assert doc["metrics"][name][str(6 * j)][v] == pytest.approx(float(r[i]), rel=5e-5)


def test_race_screen_pipelined_kernels(dev):
    """tools/stress.py in short form: the LDS-DMA pipelined kernels (counted-vmcnt hand-overs, LDS overlays) must give
    bit-identical results run after run, also with competing HBM traffic on a second stream."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "stress.py"), "12"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RACE SCREEN: CLEAN" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("K", [1056, 564, 144])
def test_split3_and_three_product_gemm(dev, K):
    """swiftk_split3: hi = bf16(v), lo = bf16(v - hi) in three column blocks ([hi | lo | hi] activations, [hi | hi | lo] weights,
    zero padding behind); the ordinary bf16 GEMM over them is an fp32-grade product (the bf16x3 engine's GEMM)."""
    from swift_amd import ops
    M, N = 512, 352
    a, w = rnd((M, K), 70), rnd((N, K), 71, 0.03)
    a3, w3 = ops.split3(a.to(dev), 0), ops.split3(w.to(dev), 1)
    ld = ops.k_pad(torch.bfloat16, 3 * K)
    assert a3.shape == (M, ld) and w3.shape == (N, ld) and a3.dtype == torch.bfloat16
    hi = a.bfloat16()
    lo = (a - hi.float()).bfloat16()
    assert torch.equal(a3[:, :K].cpu(), hi) and torch.equal(a3[:, K:2 * K].cpu(), lo) and torch.equal(a3[:, 2 * K:3 * K].cpu(), hi)
    whi = w.bfloat16()
    assert torch.equal(w3[:, :K].cpu(), whi) and torch.equal(w3[:, K:2 * K].cpu(), whi)
    assert torch.equal(w3[:, 2 * K:3 * K].cpu(), (w - whi.float()).bfloat16())
    assert float(a3[:, 3 * K:].float().abs().sum()) == 0.0 and float(w3[:, 3 * K:].float().abs().sum()) == 0.0
    assert float(((hi.float() + lo.float()) - a).abs().max() / a.abs().max()) < 2.0 ** -15
    c = ops.gemm(a3, w3, out_dtype=torch.float32)
    ref = a.double() @ w.double().t()
    e3 = float((c.cpu().double() - ref).norm() / ref.norm())
    e1 = float(((hi.double() @ whi.double().t()) - ref).norm() / ref.norm())
    print(f"K = {K}: three-product GEMM rel-L2 {e3:.2e} against fp64 (one bf16 product: {e1:.2e})")
    assert e3 < 1e-5 and e1 > 1e-3


@pytest.mark.parametrize("M,K,mlp", [(512, 1056, 2816), (1032, 564, 704), (256, 1280, 3416)])
def test_swiglu_split3_epilogue(dev, M, K, mlp):
    """SWIFTK_EPI_SWIGLU_SPLIT3: the split engine's w1 leaves silu(gate) * up as w2's operand blocks [hi | lo | hi] -- bit for bit
    what the fp32-output SwiGLU epilogue followed by swiftk_split3 leaves (same accumulators, same fp32-grade silu), pad columns
    behind the blocks untouched."""
    from swift_amd import _lib, ops
    L = _lib.lib()
    a, w = rnd((M, K), 80), rnd((2 * mlp, K), 81, 0.05)
    a3, w3 = ops.split3(a.to(dev), 0), ops.split3(w.to(dev), 1)
    h = ops.gemm(a3, w3, out_dtype=torch.float32, epilogue=_lib.EPI_SWIGLU)
    want = ops.split3(h, 0)
    ld = want.shape[1]
    got = torch.full((M, ld), 3.0, dtype=torch.bfloat16, device=dev)
    rc = L.swiftk_gemm(a3.data_ptr(), a3.stride(0), w3.data_ptr(), w3.stride(0), got.data_ptr(), ld, M, 2 * mlp, a3.shape[1], _lib.BF16,
                       _lib.BF16, _lib.EPI_SWIGLU_SPLIT3, None, None, mlp, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(got[:, :3 * mlp], want[:, :3 * mlp])
    assert ld == 3 * mlp or bool((got[:, 3 * mlp:] == 3.0).all())
    ref = torch.nn.functional.silu(a.double() @ w.double().t()[:, 0::2]) * (a.double() @ w.double().t()[:, 1::2])
    val = got[:, :mlp].float().cpu().double() + got[:, mlp:2 * mlp].float().cpu().double()
    assert float((val - ref).norm() / ref.norm()) < 2e-5


@pytest.mark.parametrize("M,N,K,pos_rows", [(2048, 1056, 576, 512), (1032, 1280, 192, 0), (512, 1536, 576, 256)])
def test_gemm_bias_pos_pair_epilogue(dev, M, N, K, pos_rows):
    """swiftk_gemm_bias_pos_pair: the patch embedding (+ bias + pos_embed) leaves as the (bf16 hi, 8-bit lo) pair -- bit for bit what
    the fp32-output SWIFTK_EPI_BIAS_POS GEMM followed by swiftk_split_pair leaves; hi's k-padding columns are not touched."""
    from swift_amd import _lib, ops
    L = _lib.lib()
    BF = torch.bfloat16
    a, w = rnd((M, K), 90).to(dev).to(BF), rnd((N, K), 91, 0.05).to(dev).to(BF)
    bias = rnd((N,), 92).to(dev)
    pos = rnd((pos_rows, N), 93).to(dev) if pos_rows else None
    x = ops.gemm(a, w, out_dtype=torch.float32, epilogue=_lib.EPI_BIAS_POS, bias=bias, pos=pos)
    ldh = ops.k_pad(BF, N) + (64 if N % 64 == 0 else 0)
    hi_ref, lo_ref = ops.split_pair(x, ldh, lo_bits=8)
    hi = torch.full((M, ldh), 5.0, dtype=BF, device=dev)
    lo = torch.zeros(M, N, dtype=torch.uint8, device=dev)
    rc = L.swiftk_gemm_bias_pos_pair(a.data_ptr(), K, w.data_ptr(), K, hi.data_ptr(), ldh, lo.data_ptr(), N, M, N, K, bias.data_ptr(),
                                     None if pos is None else pos.data_ptr(), pos_rows, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(hi[:, :N], hi_ref[:, :N]) and torch.equal(lo, lo_ref)
    assert bool((hi[:, N:] == 5.0).all())
    ref = a.float().cpu().double() @ w.float().cpu().double().t() + bias.cpu().double()
    if pos is not None:
        ref = ref + pos.cpu().double().repeat(M // pos_rows, 1)
    val = ops.pair_value(hi, lo, N).cpu().double()
    assert float((val - ref).norm() / ref.norm()) < 2e-5
