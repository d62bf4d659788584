"""-m gpu: backward-pass and loss kernels (C ABI) against torch autograd of the CPU oracle's formulas.

The training path runs bf16 GEMM operands like the reference's autocast(bfloat16) training (trainer.py:189-197), so
gradients are compared with fp32 autograd at bf16-level tolerances: per kernel on the SAME rounded inputs
(1-2e-2 relative L2, layout bugs give O(1)), and end to end by cosine similarity and relative L2 per parameter.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def rnd(shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


def s():
    return torch.cuda.current_stream().cuda_stream


def test_transpose_and_splitk_wgrad(dev):
    from swift_amd import _lib
    from swift_amd.train_engine import _transpose
    L = _lib.lib()
    M, N, K = 2048, 1056, 704  # dW[N,K] = dY^T[N,M] @ X[M,K]
    dy, x = rnd((M, N), 1).to(dev).to(BF), rnd((M, K), 2).to(dev).to(BF)
    dyt, xt = _transpose(dy, M, N), _transpose(x, M, K)
    assert torch.equal(dyt, dy.t().contiguous())
    # ragged tile edges, zero-filled padding columns [rows, ldd), and a strided (column-sliced) source
    src = rnd((1000, 300), 3).to(dev).to(BF)
    tt = _transpose(src[:, 4:284], 1000, 276, ldd=1024)   # unaligned base -> generic kernel
    assert torch.equal(tt[:, :1000], src[:, 4:280].t()) and float(tt[:, 1000:].float().abs().max()) == 0.0
    tt = _transpose(src[:, 8:288], 1000, 276, ldd=1024)   # 16-B aligned base -> vector kernel
    assert torch.equal(tt[:, :1000], src[:, 8:284].t()) and float(tt[:, 1000:].float().abs().max()) == 0.0
    ks = 4
    slabs = torch.empty(ks, N, K, dtype=torch.float32, device=dev)
    rc = L.swiftk_gemm_splitk(dyt.data_ptr(), M, xt.data_ptr(), M, slabs.data_ptr(), K, N * K, N, K, M, _lib.BF16, ks, s())
    assert rc == 0
    out = torch.full((N, K), 1.0, device=dev)
    assert L.swiftk_reduce_slabs(slabs.data_ptr(), K, N * K, ks, out.data_ptr(), K, N, K, 1, s()) == 0
    ref = dy.float().cpu().double().t() @ x.float().cpu().double() + 1.0
    assert rel_l2(out.cpu(), ref) < 1e-5
    # each slab alone is the partial product of its k-range
    ref0 = dy[: M // ks].float().cpu().double().t() @ x[: M // ks].float().cpu().double()
    assert rel_l2(slabs[0].cpu(), ref0) < 1e-5


@pytest.mark.parametrize("M,N1,N2,ks,ldp,ldq", [
    (1024, 1056, 704, 3, 1088, 704),      # partial 256-row tile (1056 = 4 x 256 + 32), garbage in the pad columns of P
    (4096, 256, 352, 16, 320, 360),       # exactly one tile, many splits, leading dimensions past the tile
    (2048, 3168, 1056, 1, 3200, 1088),    # the to_qkv gradient's shape, one k-range
    (640, 72, 1408, 2, 128, 1408),        # fewer rows than one wave tile
    (1024, 1280, 1280, 2, 1280, 1280),    # dim 1280: 320-wide tiles (even subtile count, no straddling piece)
    (1024, 512, 1536, 1, 512, 1536),      # dim 1536: 384-wide tiles, all 160 KiB of LDS
    (512, 1536, 3416, 1, 1536, 3456),     # ragged MLP width of the 468 M variant (int(8/3 1280) = 3413 -> 3416): 9 x 384, NaN pad read, never used
])
def test_tn_wgrad_equals_transposed_path(dev, M, N1, N2, ks, ldp, ldq):
    """swiftk_gemm_tn_splitk (operands token-major, transposed in LDS) against fp64 and, bit for bit, against the
    transposed-copies path it replaces."""
    from swift_amd import _lib
    from swift_amd.train_engine import _transpose
    L = _lib.lib()
    P = torch.full((M, ldp), float("nan"), dtype=BF, device=dev)   # NaN pad columns must not reach any stored value
    Q = torch.full((M, ldq), float("nan"), dtype=BF, device=dev)
    P[:, :N1] = rnd((M, N1), 5).to(dev).to(BF)
    Q[:, :N2] = rnd((M, N2), 6).to(dev).to(BF)
    slabs = torch.zeros(ks, N1, N2, dtype=torch.float32, device=dev)
    rc = L.swiftk_gemm_tn_splitk(P.data_ptr(), ldp, Q.data_ptr(), ldq, slabs.data_ptr(), N2, N1 * N2, N1, N2, M, ks, s())
    assert rc == 0
    ref = P[:, :N1].float().cpu().double().t() @ Q[:, :N2].float().cpu().double()
    assert rel_l2(slabs.sum(0).cpu(), ref) < 1e-5
    pt, qt = _transpose(P, M, N1), _transpose(Q, M, N2)
    slabs2 = torch.zeros(ks, N1, N2, dtype=torch.float32, device=dev)
    if N1 % 8 == 0 and N2 % 8 == 0:
        assert L.swiftk_gemm_splitk(pt.data_ptr(), M, qt.data_ptr(), M, slabs2.data_ptr(), N2, N1 * N2, N1, N2, M, _lib.BF16, ks, s()) == 0
        assert torch.equal(slabs, slabs2)


def test_tn_wgrad_rejects_what_it_cannot_read(dev):
    from swift_amd import _lib
    L = _lib.lib()
    P = torch.zeros(256, 1056, dtype=BF, device=dev)
    Q = torch.zeros(256, 1280, dtype=BF, device=dev)
    out = torch.zeros(1056 * 1280, device=dev)
    args = lambda ldp, ldq, n1, n2, k: (P.data_ptr(), ldp, Q.data_ptr(), ldq, out.data_ptr(), n2, n1 * n2, n1, n2, k, 1, s())
    assert L.swiftk_gemm_tn_splitk(*args(1056, 1280, 1056, 1056, 256)) == -2   # P rows end inside a 64-column block
    assert L.swiftk_gemm_tn_splitk(*args(1056, 1272, 1024, 1272, 256)) == -2   # Q rows end inside a 64-column block
    assert L.swiftk_gemm_tn_splitk(*args(1056, 1280, 1024, 1056, 200)) == -2   # token count not in 64-row k-tiles
    assert L.swiftk_gemm_tn_splitk(*args(1056, 1280, 1024, 1056, 256)) == 0


def test_embed_bwd_sums_one_pass(dev):
    """swiftk_embed_bwd_sums: bias and pos_embed gradients (both ACCUMULATE) and the bf16 copy of d(x0) in one pass, against torch
    (autograd of `patch_embed(x) + pos_embed`, swinv2.py:309-310).  The per-token sums have a fixed order: bit-reproducible."""
    from swift_amd import _lib
    L = _lib.lib()
    B, ntok, d, ld = 3, 200, 1056, 1088  # (200 tokens: a ragged last block of 16)
    dx = rnd((B * ntok, d), 81).to(dev)
    bias0, pos0 = rnd((d,), 82).to(dev), rnd((ntok, d), 83).to(dev)
    outs = []
    for _ in range(2):
        bias, pos = bias0.clone(), pos0.clone()
        dst = torch.full((B * ntok, ld), 7.0, dtype=BF, device=dev)
        _lib.check(L.swiftk_embed_bwd_sums(dx.data_ptr(), d, bias.data_ptr(), pos.data_ptr(), dst.data_ptr(), ld, B * ntok, d, ntok, s()),
                   "swiftk_embed_bwd_sums")
        outs.append((bias, pos, dst))
    bias, pos, dst = outs[0]
    assert rel_l2((bias - bias0).cpu(), dx.sum(0).cpu()) < 1e-6
    assert rel_l2((pos - pos0).cpu(), dx.view(B, ntok, d).sum(0).cpu()) < 1e-6
    assert torch.equal(dst[:, :d], dx.bfloat16()) and (dst[:, d:].float() == 0).all()
    assert torch.equal(outs[1][1], pos) and torch.equal(outs[1][2], dst)
    # without the copy
    bias2, pos2 = bias0.clone(), pos0.clone()
    _lib.check(L.swiftk_embed_bwd_sums(dx.data_ptr(), d, bias2.data_ptr(), pos2.data_ptr(), None, 0, B * ntok, d, ntok, s()), "sums")
    assert torch.equal(pos2, pos) and rel_l2(bias2.cpu(), bias.cpu()) < 1e-6
    assert L.swiftk_embed_bwd_sums(dx.data_ptr(), d, bias2.data_ptr(), pos2.data_ptr(), None, 0, B * ntok, d, 7, s()) != 0  # rows % period


@pytest.mark.parametrize("B,N,K", [(8, 50688, 1056), (5, 1056, 1056), (8, 13, 70), (2, 1, 1056)])
def test_linear_small_bwd(dev, B, N, K):
    """swiftk_linear_small_bwd (time-embedding MLP, the 2 x depth modulation Linears as one [4 depth d, d] matrix, logvar head):
    dx = dz W, dW += dz^T x, dbias += sum_b dz against torch, including row counts that are not a multiple of the kernel's
    eight rows per block and the wide n-chunks of the big matrix."""
    from swift_amd import _lib
    L = _lib.lib()
    dz, x, w = rnd((B, N), 91).to(dev), rnd((B, K), 92).to(dev), rnd((N, K), 93, 0.03).to(dev)
    dx = torch.zeros(B, K, device=dev)
    dW0, db0 = rnd((N, K), 94).to(dev), rnd((N,), 95).to(dev)
    dW, db = dW0.clone(), db0.clone()
    _lib.check(L.swiftk_linear_small_bwd(dz.data_ptr(), N, x.data_ptr(), K, w.data_ptr(), K, dx.data_ptr(), K, dW.data_ptr(), K,
                                         db.data_ptr(), B, N, K, s()), "swiftk_linear_small_bwd")
    assert rel_l2(dx.cpu(), (dz.double() @ w.double()).float().cpu()) < 1e-5
    assert rel_l2((dW - dW0).cpu(), (dz.double().t() @ x.double()).float().cpu()) < 1e-5
    assert rel_l2((db - db0).cpu(), dz.sum(0).cpu()) < 1e-5


def test_swiglu_fwd_bwd(dev):
    from swift_amd import _lib
    L = _lib.lib()
    M, mlp = 300, 2816
    h = rnd((M, 2 * mlp), 3).to(dev).to(BF)
    do = rnd((M, mlp), 4).to(dev).to(BF)
    o, dh = torch.empty(M, mlp, dtype=BF, device=dev), torch.empty(M, 2 * mlp, dtype=BF, device=dev)
    assert L.swiftk_swiglu_fwd(h.data_ptr(), 2 * mlp, o.data_ptr(), mlp, M, mlp, _lib.BF16, s()) == 0
    assert L.swiftk_swiglu_bwd(h.data_ptr(), 2 * mlp, do.data_ptr(), mlp, dh.data_ptr(), 2 * mlp, M, mlp, _lib.BF16, s()) == 0
    hc = h.float().cpu().requires_grad_(True)
    ref = F.silu(hc[:, 0::2]) * hc[:, 1::2]
    ref.backward(do.float().cpu())
    assert rel_l2(o.float().cpu(), ref.detach()) < 4e-3
    assert rel_l2(dh.float().cpu(), hc.grad) < 4e-3


@pytest.mark.parametrize("rps,d,fused", [(96, 1056, 1), (1088, 1056, 1), (1088, 1056, 0), (1536, 1536, 1), (1280, 1280, 2), (256, 1056, 1)])
def test_modnorm_bwd(dev, rps, d, fused):
    """rows_per_sample % 64 == 0 and >= d takes the one-pass kernel (tuning key 16), anything else the row pass + column pass."""
    from oracle.swinv2 import modulated_norm
    from swift_amd import _lib, ops
    L = _lib.lib()
    B = 3
    M = B * rps
    L.swiftk_set_tuning(16, fused)
    y = (rnd((M, d), 5, 2.0) + 0.3).to(dev).to(BF)
    g = rnd((M, d), 6).to(dev)
    gamma, beta = (1 + 0.1 * rnd((d,), 7)).to(dev), (0.1 * rnd((d,), 8)).to(dev)
    mod = (0.3 * rnd((B, 3 * 2 * d), 9)).to(dev)
    msl = mod[:, 2 * d: 4 * d]
    dy = torch.zeros(M, ops.k_pad(BF, d), dtype=BF, device=dev)
    dgam, dbet = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dmod = torch.zeros(B, 3 * 2 * d, device=dev)
    dsl = dmod[:, 2 * d: 4 * d]
    rc = L.swiftk_modnorm_bwd(y.data_ptr(), d, g.data_ptr(), dy.data_ptr(), dy.stride(0), gamma.data_ptr(), beta.data_ptr(),
                              msl.data_ptr(), msl.stride(0), dgam.data_ptr(), dbet.data_ptr(), dsl.data_ptr(), dsl.stride(0),
                              torch.empty(2 * M, device=dev).data_ptr(), M, d, rps, 1e-6, _lib.BF16, s())
    L.swiftk_set_tuning(16, 1)
    assert rc == 0
    yc = y.float().cpu().requires_grad_(True)
    gc, bc = gamma.cpu().requires_grad_(True), beta.cpu().requires_grad_(True)
    mc = msl.cpu().clone().requires_grad_(True)
    # modulated_norm(x [B,n,d], lat) computes the modulation Linear itself; feed an identity "Linear"
    p = {"n.norm.weight": gc, "n.norm.bias": bc}
    ln = F.layer_norm(yc.view(B, rps, d), (d,), gc, bc, 1e-6)
    out = ln * (1 + mc[:, None, :d]) + mc[:, None, d:]
    out.backward(g.cpu().view(B, rps, d))
    assert rel_l2(dy[:, :d].float().cpu(), yc.grad) < 6e-3
    assert rel_l2(dgam.cpu(), gc.grad) < 1e-4 and rel_l2(dbet.cpu(), bc.grad) < 1e-4
    assert rel_l2(dsl.cpu(), mc.grad) < 1e-4
    assert float(dmod[:, : 2 * d].abs().max()) == 0.0 and (dy.shape[1] == d or float(dy[:, d:].float().abs().max()) == 0.0)


def _prenorm(qkv, scale, heads, hd):
    B, n, _ = qkv.shape
    v = qkv.reshape(B, n, heads, 3, hd)
    tau = torch.clamp(scale, max=math.log(100.0)).exp().view(1, 1, heads, 1)
    q = v[:, :, :, 0] / v[:, :, :, 0].norm(dim=-1, keepdim=True).clamp_min(1e-12) * tau
    k = v[:, :, :, 1] / v[:, :, :, 1].norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return torch.stack([q, k, v[:, :, :, 2]], dim=3).reshape(B, n, -1)


def test_qknorm_epilogue_rn_and_bwd(dev):
    from swift_amd import _lib, ops
    L = _lib.lib()
    M, heads, hd = 512, 12, 88
    K = ops.k_pad(BF, 1056)
    a, w = rnd((M, K), 10).to(dev).to(BF), rnd((3 * heads * hd, K), 11, 0.03).to(dev).to(BF)
    scale = torch.log(torch.tensor([10.0, 3.0, 30.0, 200.0, 1.0, 10.0, 50.0, 99.0, 101.0, 5.0, 20.0, 10.0])).to(dev)
    qkvh = torch.empty(M, 3 * heads * hd, dtype=BF, device=dev)
    rn = torch.empty(M, 3 * heads, device=dev)
    rc = L.swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, qkvh.data_ptr(), 3168, M, 3168, K, _lib.BF16, _lib.BF16, _lib.EPI_QKNORM,
                       scale.data_ptr(), rn.data_ptr(), 0, s())
    assert rc == 0
    raw = (a.float().cpu().double() @ w.float().cpu().double().t()).float().requires_grad_(True)
    sc = scale.cpu().clone().requires_grad_(True)
    ref = _prenorm(raw.view(1, M, -1), sc, heads, hd)[0]
    nr = raw.detach().view(M, heads, 3, hd).norm(dim=-1)
    nr[:, :, 2] = 1.0
    assert rel_l2(rn.cpu(), (1.0 / nr).reshape(M, -1)) < 1e-5
    dq = rnd((M, 3168), 12).to(dev).to(BF)
    ref.backward(dq.float().cpu())
    dqkv = torch.zeros(M, ops.k_pad(BF, 3168), dtype=BF, device=dev)
    dscale = torch.zeros(heads, device=dev)
    # feed the kernel the exact fp32-derived normalised values rounded to bf16, as the forward would have stored them
    rc = L.swiftk_qknorm_bwd(qkvh.data_ptr(), dq.data_ptr(), 3168, rn.data_ptr(), dqkv.data_ptr(), dqkv.stride(0),
                             scale.data_ptr(), dscale.data_ptr(), M, heads, hd, _lib.BF16, s())
    assert rc == 0
    assert rel_l2(dqkv[:, :3168].float().cpu(), raw.grad) < 1.5e-2
    assert rel_l2(dscale.cpu(), sc.grad) < 2e-2
    assert float(dscale[3]) == 0.0 and float(dscale[8]) == 0.0  # heads clamped at ln(100): no gradient (swinv2.py:125)


@pytest.mark.parametrize("Mh,dim,heads", [(256, 1056, 12), (1280, 1056, 12), (128, 1280, 16), (384, 1536, 16), (256, 176, 2)])
def test_gemm_jvp_qknorm_paired_rows(dev, Mh, dim, heads):
    """swiftk_gemm_jvp(SWIFTK_EPI_QKNORM_JVP): to_qkv on [primal rows; tangent rows] with the cosine-attention prologue and its
    tangent in the epilogue (swinv2.py:121-127 under torch.func.jvp) -- against torch.func.jvp of the formula on the fp64 product,
    and against the two-launch form it replaces (swiftk_gemm + swiftk_qknorm_jvp)."""
    from swift_amd import _lib, ops
    L = _lib.lib()
    hd = dim // heads
    N, K = 3 * dim, ops.k_pad(BF, dim)
    a = torch.zeros(2 * Mh, K, dtype=BF, device=dev)
    a[:, :dim] = rnd((2 * Mh, dim), 20).to(dev).to(BF)
    w = torch.zeros(N, K, dtype=BF, device=dev)
    w[:, :dim] = rnd((N, dim), 21, 0.03).to(dev).to(BF)
    scale = (torch.log(torch.tensor([10.0, 3.0, 30.0, 200.0, 1.0, 10.0, 50.0, 99.0, 101.0, 5.0, 20.0, 10.0, 7.0, 2.0, 60.0, 40.0]))[:heads]).to(dev)
    out = torch.full((2 * Mh, N), float("nan"), dtype=BF, device=dev)
    rn = torch.full((Mh, 3 * heads), float("nan"), device=dev)
    kk = dim if dim % 64 in (0, 32) else K
    rc = L.swiftk_gemm_jvp(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, Mh, N, kk, _lib.EPI_QKNORM_JVP, scale.data_ptr(),
                           rn.data_ptr(), hd, None, 0, s())
    assert rc == 0
    torch.cuda.synchronize()
    raw = (a.float().cpu().double() @ w.float().cpu().double().t()).float()
    f = lambda z: _prenorm(z.view(1, Mh, -1), scale.cpu(), heads, hd)[0]
    ref, dref = torch.func.jvp(f, (raw[:Mh],), (raw[Mh:],))
    assert not torch.isnan(out.float()).any()
    e_p, e_t = rel_l2(out[:Mh].float().cpu(), ref), rel_l2(out[Mh:].float().cpu(), dref)
    nr = raw[:Mh].view(Mh, heads, 3, hd).norm(dim=-1)
    nr[:, :, 2] = 1.0
    assert rel_l2(rn.cpu(), (1.0 / nr).reshape(Mh, -1)) < 1e-5
    # the two-launch form on the same operands (it normalises the bf16-ROUNDED products: one rounding more)
    two = torch.empty(2 * Mh, N, dtype=BF, device=dev)
    assert L.swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, two.data_ptr(), N, 2 * Mh, N, K, _lib.BF16, _lib.BF16, _lib.EPI_NONE, None, None,
                         0, s()) == 0
    assert L.swiftk_qknorm_jvp(two.data_ptr(), two.data_ptr() + Mh * N * 2, N, scale.data_ptr(), None, Mh, heads, hd, _lib.BF16, s()) == 0
    e_p2, e_t2 = rel_l2(two[:Mh].float().cpu(), ref), rel_l2(two[Mh:].float().cpu(), dref)
    print(f"qknorm jvp epilogue Mh {Mh} dim {dim}: primal {e_p:.2e} tangent {e_t:.2e} (two launches: {e_p2:.2e} / {e_t2:.2e})")
    assert e_p < 4e-3 and e_t < 4e-3            # one bf16 rounding of exact-to-fp32 values
    assert e_p <= e_p2 * 1.05 and e_t <= e_t2 * 1.05


@pytest.mark.parametrize("Mh,dim,mlp,keep", [(256, 1056, 2816, True), (1280, 1056, 2816, False), (128, 1280, 3416, True), (384, 1536, 4096, True)])
def test_gemm_jvp_swiglu_paired_rows(dev, Mh, dim, mlp, keep):
    """swiftk_gemm_jvp(SWIFTK_EPI_SWIGLU_JVP): w1 on [primal rows; tangent rows] with silu(gate) * up and its tangent in the epilogue
    (swinv2.py:99-100 under torch.func.jvp), the primal pre-activations optionally kept for SWIFTK_EPI_SWIGLU_BWD."""
    from swift_amd import _lib, ops
    L = _lib.lib()
    N, K, kmlp = 2 * mlp, ops.k_pad(BF, dim), ops.k_pad(BF, mlp)
    a = torch.zeros(2 * Mh, K, dtype=BF, device=dev)
    a[:, :dim] = rnd((2 * Mh, dim), 30).to(dev).to(BF)
    w = torch.zeros(N, K, dtype=BF, device=dev)
    w[:, :dim] = rnd((N, dim), 31, 0.05).to(dev).to(BF)  # rows interleaved (gate_j, up_j)
    hm = torch.full((2 * Mh, kmlp), float("nan"), dtype=BF, device=dev)
    hpre = torch.full((Mh, N), float("nan"), dtype=BF, device=dev) if keep else None
    kk = dim if dim % 64 in (0, 32) else K
    rc = L.swiftk_gemm_jvp(a.data_ptr(), K, w.data_ptr(), K, None if hpre is None else hpre.data_ptr(), N, Mh, N, kk, _lib.EPI_SWIGLU_JVP,
                           None, None, 0, hm.data_ptr(), kmlp, s())
    assert rc == 0
    torch.cuda.synchronize()
    raw = (a.float().cpu().double() @ w.float().cpu().double().t()).float()
    f = lambda z: F.silu(z[:, 0::2]) * z[:, 1::2]
    ref, dref = torch.func.jvp(f, (raw[:Mh],), (raw[Mh:],))
    got = hm[:, :mlp].float().cpu()
    assert not torch.isnan(got).any()
    e_p, e_t = rel_l2(got[:Mh], ref), rel_l2(got[Mh:], dref)
    print(f"swiglu jvp epilogue Mh {Mh} dim {dim} mlp {mlp}: primal {e_p:.2e} tangent {e_t:.2e}")
    assert e_p < 4e-3 and e_t < 4e-3
    if mlp < kmlp:  # the operand's k-padding columns are the caller's (zeroed once); the kernel leaves them alone
        assert torch.isnan(hm[:, mlp:].float()).all()
    if keep:
        assert rel_l2(hpre.float().cpu(), raw[:Mh]) < 4e-3
    # the two-launch form it replaces (plain GEMM on 2 Mh rows, then swiftk_swiglu_jvp on the bf16-rounded products)
    two = torch.empty(2 * Mh, N, dtype=BF, device=dev)
    hm2 = torch.zeros(2 * Mh, kmlp, dtype=BF, device=dev)
    assert L.swiftk_gemm(a.data_ptr(), K, w.data_ptr(), K, two.data_ptr(), N, 2 * Mh, N, K, _lib.BF16, _lib.BF16, _lib.EPI_NONE, None, None,
                         0, s()) == 0
    assert L.swiftk_swiglu_jvp(two.data_ptr(), two.data_ptr() + Mh * N * 2, N, hm2.data_ptr(), hm2.data_ptr() + Mh * kmlp * 2, kmlp, Mh, mlp,
                               _lib.BF16, s()) == 0
    e_p2, e_t2 = rel_l2(hm2[:Mh, :mlp].float().cpu(), ref), rel_l2(hm2[Mh:, :mlp].float().cpu(), dref)
    print(f"  two launches: primal {e_p2:.2e} tangent {e_t2:.2e}")
    assert e_p <= e_p2 * 1.05 and e_t <= e_t2 * 1.05 and e_p2 < 1e-2 and e_t2 < 1e-2


@pytest.mark.parametrize("B", [2, 16])  # 16: 768 items on 256 persistent workgroups (buffer rotation, cross-item prefetch)
@pytest.mark.parametrize("shift", [(0, 0), (8, 8)])
@pytest.mark.parametrize("hd", [88, 80, 96])  # 80 / 96: the persistent kernel templated on head_dim (round 6)
def test_window_attention_bwd(dev, shift, B, hd):
    from oracle.swinv2 import window_token_index
    from swift_amd import _lib, ops
    L = _lib.lib()
    grid, heads = (32, 32), 12
    W3, D, LDP = 3 * heads * hd, heads * hd, 3 * heads * hd + 32
    n = grid[0] * grid[1]
    scale = torch.log(torch.tensor([10.0, 3.0, 30.0, 60.0, 1.0, 10.0, 50.0, 20.0, 15.0, 5.0, 20.0, 10.0]))
    pre = _prenorm(rnd((B, n, 3 * heads * hd), 13), scale, heads, hd).to(dev).to(BF)
    o = ops.window_attention(pre, None, grid, heads, shift, flags=_lib.ATTN_PRENORM)
    do = rnd((B, n, heads * hd), 14).to(dev).to(BF)
    dpre = torch.empty_like(pre)
    rc = L.swiftk_window_attention_bwd(pre.data_ptr(), W3, o.data_ptr(), do.data_ptr(), D, dpre.data_ptr(), B, grid[0],
                                       grid[1], heads, hd, shift[0], shift[1], _lib.BF16, s())
    assert rc == 0
    idx = window_token_index(grid, (16, 16), shift)
    pc = pre.float().cpu().requires_grad_(True)
    src = pc[:, idx.reshape(-1)].reshape(B * idx.shape[0], 256, heads, 3, hd).permute(0, 2, 1, 3, 4)
    ow = (src[..., 0, :] @ src[..., 1, :].transpose(-2, -1)).softmax(-1) @ src[..., 2, :]
    ref = torch.zeros(B, n, heads * hd).index_add(1, idx.reshape(-1), ow.permute(0, 2, 1, 3).reshape(B, n, -1))
    ref.backward(do.float().cpu())
    g = dpre.float().cpu().view(B, n, heads, 3, hd)
    gr = pc.grad.view(B, n, heads, 3, hd)
    for part, name in enumerate("qkv"):
        e = rel_l2(g[..., part, :], gr[..., part, :])
        print(f"attention bwd d{name}: rel-L2 {e:.3e}")
        assert e < 2.5e-2, name
    # the persistent kernel (default for head_dim 88) against the one-workgroup-per-item kernel it replaces: same products,
    # same bf16 roundings of P and dS up to the postponed 1/l factor
    d_old = torch.empty_like(pre)
    L.swiftk_set_tuning(9, 0)
    try:
        assert L.swiftk_window_attention_bwd(pre.data_ptr(), W3, o.data_ptr(), do.data_ptr(), D, d_old.data_ptr(), B, grid[0],
                                             grid[1], heads, hd, shift[0], shift[1], _lib.BF16, s()) == 0
    finally:
        L.swiftk_set_tuning(9, 1)
    assert rel_l2(dpre.float().cpu(), d_old.float().cpu()) < 1e-2
    # with the logit scales at hand, heads whose bound is <= 48 skip the row-maximum sweep (offset 0): same gradients
    d_s = torch.empty_like(pre)
    assert L.swiftk_window_attention_bwd_scaled(pre.data_ptr(), W3, o.data_ptr(), do.data_ptr(), D, d_s.data_ptr(), W3,
                                                scale.to(dev).data_ptr(), B, grid[0], grid[1], heads, hd, shift[0], shift[1],
                                                _lib.BF16, s()) == 0
    assert rel_l2(d_s.float().cpu(), dpre.float().cpu()) < 1e-2
    # the training engine's form: gradients written with their own row stride (the next GEMM's k-padded operand buffer) ...
    d_p = torch.full((B, n, LDP), 7.0, dtype=torch.bfloat16, device=dev)
    assert L.swiftk_window_attention_bwd_scaled(pre.data_ptr(), W3, o.data_ptr(), do.data_ptr(), D, d_p.data_ptr(), LDP,
                                                scale.to(dev).data_ptr(), B, grid[0], grid[1], heads, hd, shift[0], shift[1],
                                                _lib.BF16, s()) == 0
    assert torch.equal(d_p[..., :W3], d_s) and bool((d_p[..., W3:] == 7.0).all())
    # ... and the QK-norm backward in place on it (q-hat / k-hat vectors rewritten, v untouched) == the out-of-place call
    rn = (torch.rand(B * n, 3 * heads, device=dev) + 0.5).contiguous()
    sc = scale.to(dev).float().contiguous()
    ds_a, ds_b = torch.zeros(heads, device=dev), torch.zeros(heads, device=dev)
    out_a = torch.full((B, n, LDP), 7.0, dtype=torch.bfloat16, device=dev)
    assert L.swiftk_qknorm_bwd(pre.data_ptr(), d_s.data_ptr(), W3, rn.data_ptr(), out_a.data_ptr(), LDP, sc.data_ptr(),
                               ds_a.data_ptr(), B * n, heads, hd, _lib.BF16, s()) == 0
    assert L.swiftk_qknorm_bwd(pre.data_ptr(), d_p.data_ptr(), W3, rn.data_ptr(), d_p.data_ptr(), LDP, sc.data_ptr(),
                               ds_b.data_ptr(), B * n, heads, hd, _lib.BF16, s()) == 0
    assert torch.equal(out_a, d_p)
    assert torch.allclose(ds_a, ds_b, rtol=1e-4, atol=1e-4 * float(ds_a.abs().max()))
    for part, name in enumerate("qkv"):
        assert rel_l2(d_s.float().cpu().view(B, n, heads, 3, hd)[..., part, :], gr[..., part, :]) < 2.5e-2, name
    # both in one launch (swiftk_window_attention_bwd_qknorm: the QK-norm backward applied to the fp32 accumulators on their way
    # out) == the two-launch sequence up to the bf16 rounding of d(q-hat) / d(k-hat) it no longer performs; with one head's scale
    # above the clamp (no d(scale) there), and against the chain rule in fp32 autograd
    sc2 = sc.clone()
    sc2[3] = math.log(150.0)
    pre2 = _prenorm(rnd((B, n, 3 * heads * hd), 13), sc2.cpu(), heads, hd).to(dev).to(BF)
    o2 = ops.window_attention(pre2, None, grid, heads, shift, flags=_lib.ATTN_PRENORM)
    two = torch.full((B, n, LDP), 7.0, dtype=torch.bfloat16, device=dev)
    ds_two, ds_one = torch.zeros(heads, device=dev), torch.zeros(heads, device=dev)
    assert L.swiftk_window_attention_bwd_scaled(pre2.data_ptr(), W3, o2.data_ptr(), do.data_ptr(), D, two.data_ptr(), LDP,
                                                sc2.data_ptr(), B, grid[0], grid[1], heads, hd, shift[0], shift[1], _lib.BF16, s()) == 0
    assert L.swiftk_qknorm_bwd(pre2.data_ptr(), two.data_ptr(), W3, rn.data_ptr(), two.data_ptr(), LDP, sc2.data_ptr(),
                               ds_two.data_ptr(), B * n, heads, hd, _lib.BF16, s()) == 0
    one = torch.full((B, n, LDP), 7.0, dtype=torch.bfloat16, device=dev)
    assert L.swiftk_window_attention_bwd_qknorm(pre2.data_ptr(), W3, o2.data_ptr(), do.data_ptr(), D, one.data_ptr(), LDP,
                                                sc2.data_ptr(), rn.data_ptr(), ds_one.data_ptr(), B, grid[0], grid[1], heads, hd,
                                                shift[0], shift[1], _lib.BF16, s()) == 0
    torch.cuda.synchronize()
    assert bool((one[..., W3:] == 7.0).all())
    e12 = rel_l2(one[..., :W3].float().cpu(), two[..., :W3].float().cpu())
    eds = float((ds_one - ds_two).abs().max() / ds_two.abs().max())
    print(f"attention bwd + QK-norm bwd in one launch vs two: dqkv rel-L2 {e12:.3e}, dscale {eds:.3e}")
    assert e12 < 1e-2 and eds < 1e-2
    assert float(ds_one[3]) == 0.0 and float(ds_two[3]) == 0.0
    # fp32 chain rule: raw -> (x-hat = tau x rn, with the GIVEN rn treated as the forward's 1 / |x|) is what both kernels implement:
    #   dx = rn (tau d(x-hat) - x-hat (x-hat . d(x-hat)) / tau), with d(x-hat) from autograd of the attention core on pre2
    pc2 = pre2.float().cpu().requires_grad_(True)
    src2 = pc2[:, idx.reshape(-1)].reshape(B * idx.shape[0], 256, heads, 3, hd).permute(0, 2, 1, 3, 4)
    ow2 = (src2[..., 0, :] @ src2[..., 1, :].transpose(-2, -1)).softmax(-1) @ src2[..., 2, :]
    torch.zeros(B, n, heads * hd).index_add(1, idx.reshape(-1), ow2.permute(0, 2, 1, 3).reshape(B, n, -1)).backward(do.float().cpu())
    gh_ = pc2.grad.view(B, n, heads, 3, hd)
    xh = pc2.detach().view(B, n, heads, 3, hd)
    tau = torch.clamp(sc2.cpu(), max=math.log(100.0)).exp().view(1, 1, heads, 1)
    rnc = rn.cpu().view(B, n, heads, 3)
    want = gh_.clone()
    dotq = (xh[..., 0, :] * gh_[..., 0, :]).sum(-1, keepdim=True)
    want[..., 0, :] = rnc[..., 0:1] * (tau * gh_[..., 0, :] - xh[..., 0, :] * dotq / tau)
    dotk = (xh[..., 1, :] * gh_[..., 1, :]).sum(-1, keepdim=True)
    want[..., 1, :] = rnc[..., 1:2] * (gh_[..., 1, :] - xh[..., 1, :] * dotk)
    got = one[..., :W3].float().cpu().view(B, n, heads, 3, hd)
    for part, name in enumerate("qkv"):
        e1, e2 = rel_l2(got[..., part, :], want[..., part, :]), rel_l2(two[..., :W3].float().cpu().view(B, n, heads, 3, hd)[..., part, :], want[..., part, :])
        print(f"  d{name} vs fp32 chain rule: one launch {e1:.3e}, two launches {e2:.3e}")
        assert e1 < 2.5e-2 and e1 <= e2 * 1.1, name
    ds_want = dotq.sum(dim=(0, 1)).view(-1) * (sc2.cpu() < math.log(100.0))
    # (a sum of cancelling terms over 2k tokens; the sharp heads' d(q-hat) carry the attention kernel's bf16 roundings of P and dS)
    assert rel_l2(ds_one.cpu(), ds_want) < 5e-2 and rel_l2(ds_one.cpu(), ds_want) <= 1.2 * rel_l2(ds_two.cpu(), ds_want) + 1e-3


@pytest.mark.parametrize("shift", [(0, 0), (8, 8)])
@pytest.mark.parametrize("heads,hd", [(12, 88), (4, 80), (4, 96)])
def test_window_attention_jvp_kernel(dev, shift, heads, hd):
    """swiftk_window_attention_jvp (bf16 operands: the trainer's autocast) against torch.func.jvp of the explicit windowed
    softmax(q k^T) v (swinv2.py:129-133 with jvp=True) on the same bf16-rounded inputs."""
    from oracle.swinv2 import window_token_index
    from swift_amd import _lib
    L = _lib.lib()
    B, grid = 2, (32, 32)
    n, dq = grid[0] * grid[1], 3 * heads * hd
    scale = torch.log(torch.tensor([10.0, 3.0, 30.0, 60.0, 1.0, 10.0, 50.0, 20.0, 15.0, 5.0, 20.0, 10.0]))[:heads]
    pre = _prenorm(rnd((B, n, dq), 40), scale, heads, hd).to(dev).to(BF)
    dpre = (0.3 * rnd((B, n, dq), 41)).to(dev).to(BF)
    both = torch.cat([pre, dpre]).contiguous()  # primal rows, then tangent rows (the engine's layout)
    out = torch.full((2 * B, n, heads * hd), float("nan"), dtype=BF, device=dev)
    rc = L.swiftk_window_attention_jvp(both.data_ptr(), both.data_ptr() + B * n * dq * 2, dq, out.data_ptr(),
                                       out.data_ptr() + B * n * heads * hd * 2, heads * hd, B, grid[0], grid[1], heads, hd, shift[0],
                                       shift[1], _lib.BF16, s())
    assert rc == 0
    torch.cuda.synchronize()
    idx = window_token_index(grid, (16, 16), shift)

    def attn(z):
        src = z[:, idx.reshape(-1)].reshape(B * idx.shape[0], 256, heads, 3, hd).permute(0, 2, 1, 3, 4)
        ow = (src[..., 0, :] @ src[..., 1, :].transpose(-2, -1)).softmax(-1) @ src[..., 2, :]
        return torch.zeros(B, n, heads * hd).index_add(1, idx.reshape(-1), ow.permute(0, 2, 1, 3).reshape(B, n, -1))

    ref, dref = torch.func.jvp(attn, (pre.float().cpu(),), (dpre.float().cpu(),))
    e_o, e_d = rel_l2(out[:B].float().cpu(), ref), rel_l2(out[B:].float().cpu(), dref)
    print(f"attention tangent kernel heads {heads} hd {hd} shift {shift}: out {e_o:.3e}, tangent {e_d:.3e}")
    assert e_o < 1e-2 and e_d < 2e-2


def test_loss_kernels(dev):
    from oracle import loss as oloss
    from swift_amd import _lib
    L = _lib.lib()
    m, B, C, H, W = 2, 2, 5, 16, 32
    preds, target = rnd((m, B, C, H, W), 15), rnd((B, C, H, W), 16)
    w_var, w_lat = torch.rand(C) + 0.1, oloss.latitude_weights(H).reshape(-1)
    pd, td = preds.to(dev), target.to(dev)
    loss, dp = torch.zeros(1, device=dev), torch.empty_like(pd)
    assert L.swiftk_crps_loss(pd.data_ptr(), td.data_ptr(), w_var.to(dev).data_ptr(), w_lat.to(dev).data_ptr(), loss.data_ptr(),
                              dp.data_ptr(), m, B, C, H, W, 0.95, 1.0, s()) == 0
    pc = preds.clone().requires_grad_(True)
    ref = (w_var.view(1, C, 1, 1) * w_lat.view(1, 1, H, 1) * oloss.almost_fair_crps(pc, target, 0.95)).sum(1).mean()
    ref.backward()
    assert float(loss) == pytest.approx(float(ref), rel=1e-5)
    assert rel_l2(dp.cpu(), pc.grad) < 1e-5
    # TrigFlow
    x, z, t = rnd((B, C, H, W), 17), rnd((B, C, H, W), 18), torch.tensor([0.4, 1.3])
    Fo, lv = rnd((B, C, H, W), 19), torch.tensor([0.2, -0.3])
    xt, vt = torch.empty(B, C, H, W, device=dev), torch.empty(B, C, H, W, device=dev)
    assert L.swiftk_trigflow_prep(x.to(dev).data_ptr(), z.to(dev).data_ptr(), t.to(dev).data_ptr(), xt.data_ptr(), vt.data_ptr(),
                                  1.0, B, C * H * W, s()) == 0
    loss.zero_()
    dF, dlv = torch.empty(B, C, H, W, device=dev), torch.zeros(B, device=dev)
    assert L.swiftk_trigflow_loss(Fo.to(dev).data_ptr(), vt.data_ptr(), lv.to(dev).data_ptr(), w_var.to(dev).data_ptr(),
                                  w_lat.to(dev).data_ptr(), loss.data_ptr(), dF.data_ptr(), dlv.data_ptr(), 1.0, B, C, H, W, 1.0,
                                  s()) == 0
    Fc, lc = Fo.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    net = lambda xx, tt, c, a, return_logvar=False: (Fc, lc)
    tau = torch.tan(t).view(B, 1, 1, 1)
    ref = oloss.trigflow_loss(net, x, tau, z, w_var.view(1, C, 1, 1), w_lat.view(1, 1, H, 1), 1.0, return_logvar=True)
    ref.backward()
    assert float(loss) == pytest.approx(float(ref), rel=1e-5)
    assert rel_l2(dF.cpu(), Fc.grad) < 1e-5 and rel_l2(dlv.cpu(), lc.grad) < 1e-4
    c, sn = torch.cos(t).view(B, 1, 1, 1), torch.sin(t).view(B, 1, 1, 1)
    assert rel_l2(xt.cpu(), c * x + sn * z) < 1e-6 and rel_l2(vt.cpu(), c * z - sn * x) < 1e-6


SMALLB = dict(img=(64, 64), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1056, heads=12, depth=2)


def test_train_engine_gradients_vs_oracle_autograd(dev):
    from oracle.swinv2 import OracleNet, SwinCfg
    from swift_amd.models.precond import PassPrecond
    from swift_amd.train_engine import SwinTrainEngine
    from swift_amd.utils.detinit import det_normal, swinv2_state
    c, seed = SMALLB, 21
    nv, nf = c["n_vars"], c["n_forc"]
    mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2],
                depth=c["depth"], dim=c["dim"], heads=c["heads"], logvar=True)
    net = PassPrecond(mcfg, img_resolution=list(c["img"]), img_channels=nv, condition_channels=nv + nf, auxiliary_dim=1)
    state = swinv2_state(grid=(32, 32), in_channels=2 * nv + nf, out_channels=nv, patch_size=(2, 2), depth=c["depth"],
                         dim=c["dim"], heads=c["heads"], logvar=True, seed=seed)
    for k in state:  # no head clamped at ln(100) here: keep every scale gradient alive
        if k.endswith(".scale"):
            state[k] = state[k].clamp(max=4.0)
    net.load_state_dict(state)
    net = net.to(dev)
    B = 2
    x, cond = det_normal((B, nv, 64, 64), seed, "x"), det_normal((B, nv + nf, 64, 64), seed, "cond")
    t, aux = torch.tensor([0.5, 1.4]), torch.tensor([[0.6], [1.2]])
    R, rl = det_normal((B, nv, 64, 64), seed, "R"), torch.tensor([0.7, -0.4])
    eng = SwinTrainEngine(net.model)
    out, lv, ctx = eng.forward([x.to(dev), cond[:, :nv].to(dev), cond[:, nv:].to(dev)], [1.0, 1.0, 1.0], t.to(dev), aux.to(dev),
                               want_logvar=True)
    dins = eng.backward(ctx, R.to(dev), rl.to(dev), need_input_grad=(True, True, False))
    torch.cuda.synchronize()
    # oracle: fp32 autograd on the CPU
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    ocfg = SwinCfg(img_resolution=c["img"], in_channels=2 * nv + nf, out_channels=nv, window_size=(16, 16), shift_size=(8, 8),
                   patch_size=(2, 2), depth=c["depth"], dim=c["dim"], heads=c["heads"], auxiliary_dim=1, logvar=True)
    onet = OracleNet(ocfg, st, nv, nv + nf)
    xo, co = x.clone().requires_grad_(True), cond.clone().requires_grad_(True)
    yo, lvo = onet(xo, t, co, aux, return_logvar=True)
    ((yo * R).sum() + (lvo * rl).sum()).backward()
    e_out = rel_l2(out.cpu(), yo.detach())
    print(f"train-engine forward vs oracle: rel-L2 {e_out:.3e}")
    assert e_out < 1e-1 and rel_l2(lv.cpu(), lvo.detach()) < 1e-3
    worst = 0.0
    named = dict(net.named_parameters())
    for k, p in named.items():
        g, gr = p.grad.float().cpu().flatten().double(), st[k].grad.flatten().double()
        cos = float((g @ gr) / (g.norm() * gr.norm()).clamp_min(1e-30))
        e = rel_l2(g, gr)
        worst = max(worst, e)
        print(f"{k:55s} cos {cos:.4f} rel-L2 {e:.3e}")
        assert cos > 0.99, (k, cos, e)  # measured: >= 0.994 for every one of the 40 tensors
    for gi, ref in ((dins[0], xo.grad), (dins[1], co.grad[:, :nv])):
        gg, rr = gi.cpu().flatten().double(), ref.flatten().double()
        cos = float((gg @ rr) / (gg.norm() * rr.norm()))
        print(f"input grad: cos {cos:.4f} rel-L2 {rel_l2(gg, rr):.3e}")
        assert cos > 0.99
    assert dins[2] is None


def _build_pair(dev, seed, logvar=False, depth=2, dim=None, heads=None):
    from oracle.swinv2 import OracleNet, SwinCfg
    from swift_amd.models.precond import PassPrecond
    from swift_amd.utils.detinit import swinv2_state
    c = dict(SMALLB, depth=depth)
    if dim is not None:
        c.update(dim=dim, heads=heads)
    nv, nf = c["n_vars"], c["n_forc"]
    mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2],
                depth=depth, dim=c["dim"], heads=c["heads"], logvar=logvar)
    net = PassPrecond(mcfg, img_resolution=list(c["img"]), img_channels=nv, condition_channels=nv + nf, auxiliary_dim=1)
    state = swinv2_state(grid=(32, 32), in_channels=2 * nv + nf, out_channels=nv, patch_size=(2, 2), depth=depth, dim=c["dim"],
                         heads=c["heads"], logvar=logvar, seed=seed)
    for k in state:
        if k.endswith(".scale"):
            state[k] = state[k].clamp(max=3.0)
    net.load_state_dict(state)
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    ocfg = SwinCfg(img_resolution=c["img"], in_channels=2 * nv + nf, out_channels=nv, window_size=(16, 16), shift_size=(8, 8),
                   patch_size=(2, 2), depth=depth, dim=c["dim"], heads=c["heads"], auxiliary_dim=1, logvar=logvar)
    return net.to(dev), OracleNet(ocfg, st, nv, nv + nf), st


def _grad_report(net, st, floor=0.99):
    worst = 1.0
    for k, p in net.named_parameters():
        g, gr = p.grad.float().cpu().flatten().double(), st[k].grad.flatten().double()
        cos = float((g @ gr) / (g.norm() * gr.norm()).clamp_min(1e-30))
        worst = min(worst, cos)
        assert cos > floor, (k, cos)
    return worst


def _dataset(seed):
    from swift_amd.data.era5 import SyntheticERA5Dataset
    names = ["2m_temperature", "10m_u_component_of_wind", "10m_v_component_of_wind", "mean_sea_level_pressure"]
    for v in ["geopotential", "u_component_of_wind", "v_component_of_wind", "temperature", "specific_humidity"]:
        names += [f"{v}_{l}" for l in [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]]
    return SyntheticERA5Dataset(names, ["f0", "f1", "f2"], img_resolution=(64, 64), length=40, seed=seed, random_stats=True)


def test_trigflow_loss_and_grads_vs_oracle(dev):
    from oracle import loss as oloss
    from swift_amd.training.loss import TrigFlowLoss
    from swift_amd.utils.detinit import det_normal
    net, onet, st = _build_pair(dev, 31, logvar=True)
    ds = _dataset(31)
    L = TrigFlowLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), sigma_data=1.0).to(dev)
    B = 2
    x, cond, z = det_normal((B, 69, 64, 64), 31, "x"), det_normal((B, 72, 64, 64), 31, "c"), det_normal((B, 69, 64, 64), 31, "z")
    tau, aux = torch.tensor([0.3, 4.0]).view(B, 1, 1, 1), torch.tensor([0.6, 0.6])
    from swift_amd.training.trainer import GradAllReduce
    ddp = GradAllReduce(net)
    ddp.zero_grad_flat()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = L(ddp, x.to(dev), condition=cond.to(dev), auxiliary=aux.to(dev), _tau=tau.to(dev), _z=z.to(dev))
    loss.backward()
    ref = oloss.trigflow_loss(onet, x, tau, z, L.w_var.cpu(), L.w_lat.cpu(), 1.0, condition=cond, auxiliary=aux, return_logvar=True)
    ref.backward()
    print(f"trigflow loss {float(loss):.6f} vs oracle {float(ref):.6f}; worst grad cosine {_grad_report(net, st):.4f}")
    assert float(loss) == pytest.approx(float(ref), rel=1e-3)  # measured 6e-5 (bf16 operands vs fp32 autograd)


@pytest.mark.parametrize("steps,keep", [(3, "0"), (3, None), (4, "3"), (4, "0")])  # 4 = BASELINE configs[4] (finetune/multistep.yaml's last interval)
def test_crps_multistep_loss_and_grads_vs_oracle(dev, steps, keep, monkeypatch):
    """keep: rollout steps whose activations stay resident for the backward walk ("0": every step recomputed as in the
    reference's checkpoint_sequential; None: the memory rule, which keeps all 2 x steps calls of this small net; "3": mixed)."""
    from oracle import loss as oloss
    if keep is None:
        monkeypatch.delenv("SWIFTK_CRPS_KEEP", raising=False)
    else:
        monkeypatch.setenv("SWIFTK_CRPS_KEEP", keep)
    from oracle.rollout import Stats
    from swift_amd.training.loss import CRPSLoss
    from swift_amd.training.trainer import GradAllReduce
    from swift_amd.utils.detinit import det_normal
    net, onet, st = _build_pair(dev, 32)
    ds = _dataset(32)
    L = CRPSLoss(ds, sigma_data=1.0, ensemble_size=2, alpha=0.95).to(dev)
    B = 2
    target, cond = det_normal((B, 69, 64, 64), 32, "t"), det_normal((B, 72, 64, 64), 32, "c")
    aux, idx = torch.tensor([0.6, 0.6]), [0, 4]
    lat = [[det_normal((B, 69, 64, 64), 32, f"l{e}{i}") for i in range(steps)] for e in range(2)]
    ddp = GradAllReduce(net)
    ddp.zero_grad_flat()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = L(ddp, target.to(dev), condition=cond.to(dev), auxiliary=aux.to(dev), idx=idx, steps=steps, _latents=lat)
    loss.backward()
    stats = Stats(ds.x_means, ds.x_stds, {6: ds.t_stds[6]}, n_vars=69, n_forc=3)
    forc = lambda i: torch.stack([ds.get_forcings(j + i) for j in idx], 0)
    ref = oloss.crps_multistep_loss(onet, stats, target, cond, aux, forc, lat, L.w_var.cpu(), L.w_lat.cpu(), steps=steps, alpha=0.95)
    ref.backward()
    print(f"CRPS(steps={steps}, keep={keep}) loss {float(loss):.6f} vs oracle {float(ref):.6f}; worst grad cosine {_grad_report(net, st, 0.999):.4f}")
    assert float(loss) == pytest.approx(float(ref), rel=2e-3)  # measured 4.4e-4 at steps = 3
    assert L.last_n_keep == (2 * steps if keep is None else int(keep))


def test_network_tangent_vs_oracle_jvp(dev):
    """jvp_engine.SwinJvpEngine (explicit tangent kernels) against torch.func.jvp of the CPU oracle with jvp=True
    (what loss.py:212-220 does with the reference network): fp32 operands tight, bf16 operands at bf16 noise."""
    from swift_amd.jvp_engine import SwinJvpEngine
    from swift_amd.models.precond import _process_auxiliary
    from swift_amd.utils.detinit import det_normal
    net, onet, _ = _build_pair(dev, 41)
    B = 2
    x, cond = det_normal((B, 69, 64, 64), 41, "x"), det_normal((B, 72, 64, 64), 41, "c")
    vx = det_normal((B, 69, 64, 64), 41, "vx")
    t, vt, aux = torch.tensor([0.4, 1.3]), torch.tensor([0.35, 0.2]), torch.tensor([0.6, 0.6])
    with torch.no_grad():
        f = lambda xx, tt: onet(xx, tt, cond, aux, jvp=True)
        Fref, dref = torch.func.jvp(f, (x, t), (vx, vt))
    auxd = _process_auxiliary(aux.to(dev), 1, B, dev)
    errs = {}
    for dt in (torch.float32, torch.bfloat16):
        eng = SwinJvpEngine(net.model, dt)
        dF = eng.jvp([x.to(dev), cond.to(dev)], vx.to(dev), t.to(dev), vt.to(dev), auxd)
        assert torch.isfinite(dF).all()
        errs[dt] = rel_l2(dF.cpu(), dref)
    print(f"network tangent vs oracle jvp: fp32 rel-L2 {errs[torch.float32]:.3e}, bf16 rel-L2 {errs[torch.bfloat16]:.3e}")
    assert errs[torch.float32] < 1e-4
    assert errs[torch.bfloat16] < 8e-2


def test_scm_loss_and_grads_vs_oracle(dev):
    from oracle import loss as oloss
    from swift_amd.training.loss import SCMLoss
    from swift_amd.training.trainer import GradAllReduce
    from swift_amd.utils.detinit import det_normal
    net, onet, st = _build_pair(dev, 42, logvar=True)
    ds = _dataset(42)
    L = SCMLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), sigma_data=1.0, tangent_warmup_kimg=3,
                jvp_dtype="f32").to(dev)
    B = 2
    x, cond, z = det_normal((B, 69, 64, 64), 42, "x"), det_normal((B, 72, 64, 64), 42, "c"), det_normal((B, 69, 64, 64), 42, "z")
    tau, aux = torch.tensor([0.3, 4.0]).view(B, 1, 1, 1), torch.tensor([0.6, 0.6])
    ddp = GradAllReduce(net)
    ddp.zero_grad_flat()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = L(ddp, x.to(dev), 1200, condition=cond.to(dev), auxiliary=aux.to(dev), _tau=tau.to(dev), _z=z.to(dev))
    loss.backward()
    ref = oloss.scm_loss(onet, x, tau, z, L.w_var.cpu(), L.w_lat.cpu(), step=1200, sigma_data=1.0, tangent_warmup_kimg=3,
                         condition=cond, auxiliary=aux, return_logvar=True)
    ref.backward()
    print(f"sCM loss {float(loss):.6f} vs oracle {float(ref):.6f}; worst grad cosine {_grad_report(net, st, 0.999):.4f}")
    assert float(loss) == pytest.approx(float(ref), rel=1e-3)  # measured 6e-6
    # bf16 tangent pass (the trainer's autocast): same loss within bf16 noise
    L.jvp_dtype = torch.bfloat16
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lb = L(ddp, x.to(dev), 1200, condition=cond.to(dev), auxiliary=aux.to(dev), _tau=tau.to(dev), _z=z.to(dev))
    assert float(lb) == pytest.approx(float(ref), rel=5e-2)
    # ... and that pass is ALSO the forward pass of the backward (one network pass instead of the reference's two): its
    # primal rows, kept per layer, give the same parameter gradients
    ddp.zero_grad_flat()
    lb.backward()
    print(f"sCM one-pass (bf16 tangent rows = saved activations): loss {float(lb):.6f}; worst grad cosine {_grad_report(net, st, 0.999):.4f}")


def test_scm_distillation_loss_and_grads_vs_oracle(dev):
    """SCMLoss(distillation=True): dx_t/dt from a frozen v-prediction teacher (training/loss.py:204-208; the oracle's
    ``teacher=`` branch is pinned to the reference by tests/golden/scm_distill_tiny.npz).  The teacher runs on the fp32
    inference engine here (no autocast) so that the comparison is at the fp32-tangent level; under the trainer's bf16
    autocast it runs on the bf16 engine like the reference's teacher under its autocast."""
    from oracle import loss as oloss
    from swift_amd.training.loss import SCMLoss
    from swift_amd.training.trainer import GradAllReduce
    from swift_amd.utils.detinit import det_normal
    net, onet, st = _build_pair(dev, 43, logvar=True)
    teacher, oteacher, _ = _build_pair(dev, 53, logvar=False)
    teacher.eval().requires_grad_(False)
    ds = _dataset(43)
    L = SCMLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), sigma_data=1.0, tangent_warmup_kimg=3,
                distillation=True, jvp_dtype="f32").to(dev)
    B = 2
    x, cond, z = det_normal((B, 69, 64, 64), 43, "x"), det_normal((B, 72, 64, 64), 43, "c"), det_normal((B, 69, 64, 64), 43, "z")
    tau, aux = torch.tensor([0.3, 4.0]).view(B, 1, 1, 1), torch.tensor([0.6, 0.6])
    ddp = GradAllReduce(net)
    ddp.zero_grad_flat()
    loss = L(ddp, x.to(dev), 1200, condition=cond.to(dev), auxiliary=aux.to(dev), net_pretrained=teacher, _tau=tau.to(dev),
             _z=z.to(dev))
    loss.backward()
    kw = dict(step=1200, sigma_data=1.0, tangent_warmup_kimg=3, condition=cond, auxiliary=aux, return_logvar=True)
    ref = oloss.scm_loss(onet, x, tau, z, L.w_var.cpu(), L.w_lat.cpu(), teacher=oteacher, **kw)
    ref.backward()
    with torch.no_grad():
        plain = oloss.scm_loss(onet, x, tau, z, L.w_var.cpu(), L.w_lat.cpu(), **kw)
    print(f"sCM distillation loss {float(loss):.6f} vs oracle {float(ref):.6f} (without teacher {float(plain):.6f}); "
          f"worst grad cosine {_grad_report(net, st, 0.999):.4f}")
    assert float(loss) == pytest.approx(float(ref), rel=1e-3)
    assert abs(float(plain) - float(ref)) > 10 * abs(float(loss) - float(ref))  # the teacher is what is being tested
    assert all(p.grad is None for p in teacher.parameters())
    # the trainer's configuration: bf16 autocast around the loss (teacher on the bf16 engine, bf16 tangent pass)
    L.jvp_dtype = torch.bfloat16
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lb = L(ddp, x.to(dev), 1200, condition=cond.to(dev), auxiliary=aux.to(dev), net_pretrained=teacher, _tau=tau.to(dev),
               _z=z.to(dev))
    assert float(lb) == pytest.approx(float(ref), rel=5e-2)


def test_muon_with_aux_adam_vs_reference_golden(dev):
    """swift_amd MuonWithAuxAdam (Newton-Schulz products on swiftk_gemm) against two steps of the reference's
    SingleDeviceMuonWithAuxAdam (tests/golden/muon_tiny.npz).  The orthogonalisation runs in bf16 in both; a different
    summation order inside the bf16 GEMMs moves the update by bf16 noise, amplified by 5 quintic iterations."""
    import numpy as np
    from conftest import load_golden
    from swift_amd.training.optimizers.muon import MuonWithAuxAdam, zeropower_via_newtonschulz5
    from oracle import muon as omuon
    g = load_golden("muon_tiny")
    params = [torch.nn.Parameter(torch.from_numpy(g[f"p{i}_0"]).to(dev)) for i in range(5)]
    opt = MuonWithAuxAdam([dict(params=params[:3], use_muon=True, lr=0.02, weight_decay=0.01),
                           dict(params=params[3:], use_muon=False, lr=3e-4, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-10)])
    for st in range(2):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(g[f"g{i}_{st}"]).to(dev)
        opt.step()
        for i, p in enumerate(params):
            # compare the STEP taken (p_new - p_old of the golden trajectory is ~4 % of |p| for the Muon groups)
            mine = p.detach().cpu() - torch.from_numpy(g[f"p{i}_{st}"])
            ref = torch.from_numpy(g[f"p{i}_{st + 1}"] - g[f"p{i}_{st}"])
            err = rel_l2(mine, ref)
            print(f"step {st} param {i} {tuple(p.shape)}: update rel-L2 {err:.3e}")
            assert err < (8e-2 if i < 3 else 1e-4), (st, i, err)
            p.data.copy_(torch.from_numpy(g[f"p{i}_{st + 1}"]).to(dev))  # continue from the golden trajectory
    # the orthogonaliser alone at a Swift-B shape (wo: 1056 x 1056) against the CPU restatement
    G = rnd((1056, 1056), 55)
    X = zeropower_via_newtonschulz5(G.to(dev), 5).float().cpu()
    Xr = omuon.newton_schulz5(G, 5).float()
    e = rel_l2(X, Xr)
    sv = torch.linalg.svdvals(X.double())
    print(f"Newton-Schulz 1056x1056: rel-L2 vs CPU restatement {e:.3e}; singular values in [{float(sv.min()):.3f}, {float(sv.max()):.3f}]")
    assert e < 3e-2 and float(sv.max()) < 1.7 and float((sv > 0.5).double().mean()) > 0.8  # most of the spectrum pushed to ~1


def test_graph_capture_survives_other_threads_using_the_runtime(dev):
    """A real training run captures its launch sequences while the DataLoader's pin-memory thread allocates pinned batches and
    copies them to the device; under torch's default capture mode ("global") such a call from ANOTHER thread invalidates the
    capture (hipErrorStreamCaptureInvalidated in the first captured pass of `python -m swift_amd.train` at Swift-B).
    ``graphs.capture`` polices the capturing thread only."""
    import threading
    from swift_amd import graphs, ops
    stop = threading.Event()
    seen = []

    def pinner():  # what torch.utils.data's pin_memory thread does, as fast as it can
        side = torch.cuda.Stream(device=dev)
        while not stop.is_set():
            h = torch.empty(1 << 18).pin_memory()
            with torch.cuda.stream(side):
                seen.append(h.to(dev, non_blocking=True).sum())
            side.synchronize()
            del seen[:-4]

    th = threading.Thread(target=pinner, daemon=True)
    th.start()
    try:
        a, w = rnd((512, 1088), 3).to(dev).to(BF), rnd((352, 1088), 4).to(dev).to(BF)
        for rep in range(12):
            cache = graphs.GraphCache()
            fn = lambda x: (ops.gemm(x, w) * 2.0,)
            ref = fn(a)[0].clone()
            for _ in range(3):  # eager, capture + replay, replay
                out = cache.call(("k", rep), fn, [a])[0]
            assert cache._graphs and torch.equal(out, ref)
    finally:
        stop.set()
        th.join(timeout=20)


def test_muon_stacked_step_vs_matrix_by_matrix(dev):
    """One rank: MuonWithAuxAdam orthogonalises the same-shape matrices of a group TOGETHER (swiftk_gemm_batched, multi-tensor
    momentum / update launches).  Against the matrix-by-matrix update on a copy: the same step up to bf16 noise through the five
    quintic iterations (a different GEMM kernel sums in a different order), and the batched GEMM itself against fp64."""
    from swift_amd import _lib
    from swift_amd.training.optimizers import muon as pm
    L = _lib.lib()
    a = (rnd((3, 320, 192), 90) * 0.3).to(dev).to(BF)
    w = (rnd((3, 96, 192), 91) * 0.3).to(dev).to(BF)
    out = torch.zeros(3, 320, 104, dtype=BF, device=dev)
    assert L.swiftk_gemm_batched(a.data_ptr(), 192, 320 * 192, w.data_ptr(), 192, 96 * 192, out.data_ptr(), 104, 320 * 104, 3, 320, 96,
                                 192, _lib.BF16, _lib.BF16, s()) == 0
    ref = torch.einsum("lmk,lnk->lmn", a.double().cpu(), w.double().cpu())
    assert rel_l2(out[:, :, :96].double().cpu(), ref) < 4e-3 and float(out[:, :, 96:].float().abs().max()) == 0.0
    # (128, 1024): the matrix-by-matrix path sums such a contraction over split-K slabs, the stacked one in a single k-loop;
    # the small ones take the same arithmetic order in both (bit-equal)
    shapes = [(64, 160)] * 3 + [(128, 1024)] * 2 + [(96, 32)] * 2 + [(1, 12, 1, 1), (48, 48)]

    def make():
        ps = [torch.nn.Parameter((rnd(sh, 80 + i) * 0.05).to(dev)) for i, sh in enumerate(shapes)]
        return ps, pm.MuonWithAuxAdam([dict(params=ps, use_muon=True, lr=0.02, weight_decay=0.01)])

    def run(stacked):
        ps, opt = make()
        if not stacked:
            opt._muon_group_stacked = lambda group: opt._muon_group(group, 1, 0, False)
        for st in range(3):
            for i, q in enumerate(ps):
                q.grad = rnd(q.shape, 900 + 10 * st + i).to(dev)
            before = [q.detach().clone() for q in ps]
            opt.step()
        return [q.detach() - b for q, b in zip(ps, before)]  # the third step taken

    for i, (u, v) in enumerate(zip(run(True), run(False))):
        e = rel_l2(u.float().cpu(), v.float().cpu())
        print(f"param {i} {tuple(u.shape)}: stacked vs matrix-by-matrix step rel-L2 {e:.2e}")
        assert e < 5e-2


def test_trainer_trajectory_vs_oracle_net_and_oracle_trainer(dev, tmp_path, monkeypatch):
    """Three optimisation steps of ``Trainer.train_step`` (multistep-CRPS loss, bf16 autocast, fused AdamW + EMA kernel,
    LR warm-up then cosine) against the CPU oracle, step by step:
      (a) loss and every parameter gradient vs fp32 autograd of ``oracle.loss.crps_multistep_loss`` on the oracle net
          holding the trainer's CURRENT weights (trainer.py:189-197);
      (b) new weights, EMA weights and learning rates vs ``oracle.trainer.OracleTrainerState.step`` (pinned to the reference's
          ``Trainer._backward_step``, trainer.py:199-247) fed the trainer's own gradients;
      (c) a free-running oracle trajectory (its own gradients, its own optimiser state): losses stay together and the net
          displacement of every large tensor points the same way.
    Then six more steps with ``profile=True`` bookkeeping (trainer.py:155-177) and the checkpoint format."""
    import json
    from oracle import loss as oloss
    from oracle.rollout import Stats
    from oracle.trainer import OracleTrainerState
    from swift_amd.training.loss import CRPSLoss
    from swift_amd.training.trainer import Trainer
    from swift_amd.train import adamw_param_groups
    from swift_amd.utils.detinit import det_normal
    monkeypatch.chdir(tmp_path)
    net, onet, st = _build_pair(dev, 33)
    net.train().requires_grad_(True)
    ds = _dataset(33)
    opt = torch.optim.AdamW(adamw_param_groups(net, 1e-5), lr=2e-4, betas=(0.9, 0.95), eps=1e-6)
    assert len(opt.param_groups[1]["params"]) == 1 + 2 * 2 * 2  # pos_embed + LayerNorm affine of 2 layers x 2 norms
    B, E, nsteps = 2, 2, 9
    lats = [[[det_normal((B, 69, 64, 64), 33, f"l{k}{e}")] for e in range(E)] for k in range(nsteps)]

    class FixedNoiseCRPS(CRPSLoss):  # the k-th call draws the k-th prepared latents (the oracle gets the same ones)
        calls = 0

        def forward(self, *a, **kw):
            kw["_latents"] = lats[self.calls]
            self.calls += 1
            return super().forward(*a, **kw)

    sched = dict(lr_rampup_kimg=0.004, lr_min_factor=0.1, lr_cosine_anneal=True, total_kimg=0.03)
    tr = Trainer(net, opt, FixedNoiseCRPS(ds, 1.0, E, 0.95).to(dev), ema_halflife_kimg=1, ema_rampup_ratio=0.05, kimg_per_tick=1,
                 checkpoint_ticks=None, device=dev, profile=True, **sched)
    tr.global_batch_size = B
    x = det_normal((B, 72, 64, 64), 33, "c").to(dev)
    t = (0.5 * det_normal((B, 69, 64, 64), 33, "t")).to(dev)
    delta, idx = torch.tensor([0.6, 0.6]), [0, 3]
    names = [n for n, _ in net.named_parameters()]
    group_of = [1 if ("pos_embed" in n or ("norm" in n and "modulation" not in n)) else 0 for n in names]
    mk_state = lambda: OracleTrainerState([p.detach().cpu() for p in net.parameters()], [p.detach().cpu() for p in tr.ema.parameters()],
                                          group_of, base_lr=[2e-4, 2e-4], weight_decay=[1e-5, 0.0], betas=(0.9, 0.95), eps=1e-6)
    fed, free = mk_state(), mk_state()          # (b): fed the trainer's gradients; (c): free-running
    p_start = [p.detach().cpu().clone() for p in net.parameters()]
    stats = Stats(ds.x_means, ds.x_stds, {6: ds.t_stds[6]}, n_vars=69, n_forc=3)
    forc = lambda i: torch.stack([ds.get_forcings(j + i) for j in idx], 0)
    w_var, w_lat = tr.loss_fn.w_var.cpu(), tr.loss_fn.w_lat.cpu()
    tkw = dict(global_batch_size=B, ema_halflife_kimg=1, ema_rampup_ratio=0.05, **sched)

    def oracle_loss_and_grads(weights, k):
        with torch.no_grad():
            for n, w in zip(names, weights):
                st[n].copy_(w)
        for v in st.values():
            v.grad = None
        val = oloss.crps_multistep_loss(onet, stats, t.cpu(), x.cpu(), delta, forc, lats[k], w_var, w_lat, steps=1, alpha=0.95)
        val.backward()
        return float(val), [st[n].grad.clone() for n in names]

    snap = {}
    inner = tr._optimizer_step
    tr._optimizer_step = lambda nimg, flat: (snap.__setitem__("g", [p.grad.detach().cpu().clone() for p in net.parameters()]),
                                             inner(nimg, flat))[1]
    losses, free_losses = [], []
    for k in range(3):
        nimg = B * (k + 1)
        w_now = [p.detach().cpu().clone() for p in net.parameters()]
        loss = float(tr.train_step(x, t, idx, delta, global_nimg=nimg, steps=1))
        # (a) the trainer's loss and gradients at w_now vs the oracle's
        ref, og = oracle_loss_and_grads(w_now, k)
        assert loss == pytest.approx(ref, rel=2e-3), k
        for n, g, r in zip(names, snap["g"], og):
            cos = float((g.flatten().double() @ r.flatten().double()) / (g.norm().double() * r.norm().double()).clamp_min(1e-30))
            assert cos > 0.999, (k, n, cos)
        # (b) the optimisation step itself, on the trainer's gradients
        lr = fed.step(snap["g"], nimg, **tkw)
        assert [gr["lr"] for gr in opt.param_groups] == pytest.approx(lr, rel=1e-6)
        for n, p, e, rp, re in zip(names, net.parameters(), tr.ema.parameters(), fed.p, fed.e):
            assert rel_l2(p.detach().cpu(), rp) < 1e-5 and rel_l2(e.detach().cpu(), re) < 1e-5, (k, n)
        # (c) free-running oracle
        fl, fg = oracle_loss_and_grads(free.p, k)
        free.step(fg, nimg, **tkw)
        losses.append(loss)
        free_losses.append(fl)
    print("CRPS finetune losses:", [f"{v:.5f}" for v in losses], "free-running oracle:", [f"{v:.5f}" for v in free_losses])
    assert losses == pytest.approx(free_losses, rel=5e-3)
    for n, p, p0, q in zip(names, net.parameters(), p_start, free.p):
        if p.numel() >= 1 << 16:  # displacement after three Adam steps (bf16 noise flips the sign of near-zero gradients)
            d1, d2 = (p.detach().cpu() - p0).flatten().double(), (q - p0).flatten().double()
            assert float(d1 @ d2 / (d1.norm() * d2.norm())) > 0.9, n
    assert lr[0] < 2e-4 and opt.param_groups[0]["lr"] < 2e-4  # past the warm-up, on the cosine
    # ---- profile=True: wait 2 / warm-up 2 / active 5 iterations, then the record (trainer.py:155-177, 389-396)
    assert not tr.prof.done
    tr._optimizer_step = inner  # (the gradient snapshots above copied to the host inside the "optimizer" phase)
    for k in range(3, nsteps):
        assert math.isfinite(float(tr.train_step(x, t, idx, delta, global_nimg=B * (k + 1), steps=1)))
    assert tr.prof.done and set(tr.prof.summary) == {"forward", "backward", "allreduce", "optimizer"}
    rec = json.load(open(tmp_path / "rank0_prof.json"))
    assert len(rec["traceEvents"]) == 5 * 4 and all(e["dur"] > 0 for e in rec["traceEvents"])
    assert rec["otherData"]["mean_ms_per_phase"]["backward"] > rec["otherData"]["mean_ms_per_phase"]["optimizer"]
    tr._save_checkpoint(3000)
    state = torch.load(tmp_path / "checkpoints" / "checkpoint-000003.pt", weights_only=True)
    assert set(state) == {"ema", "net", "optimizer", "scaler"} and len(state["ema"]) == len(net.state_dict())


@pytest.mark.parametrize("kind", ["crps", "scm"])
def test_training_loop_with_graph_replay_tracks_the_eager_loop(dev, tmp_path, monkeypatch, kind):
    """Six optimiser steps with the launch sequences replayed as HIP graphs (steps 3+ are pure replays, with weight updates,
    resident rollout activations / the one-pass sCM buffers in between) against the same six steps issued eagerly: the
    loss trajectories agree to the noise of the fp32 atomics amplified by Adam (a stale buffer or a mis-replayed sequence
    shows up as a diverging or non-finite loss by step 3)."""
    from swift_amd.training.loss import CRPSLoss, SCMLoss
    from swift_amd.training.trainer import Trainer
    from swift_amd.train import adamw_param_groups
    from swift_amd.utils.detinit import det_normal
    monkeypatch.chdir(tmp_path)

    def run(graph_mode):
        monkeypatch.setenv("SWIFTK_TRAIN_GRAPHS", graph_mode)
        net, _, _ = _build_pair(dev, 35, logvar=(kind == "scm"))
        net.train().requires_grad_(True)
        ds = _dataset(35)
        opt = torch.optim.AdamW(adamw_param_groups(net, 1e-5), lr=1e-4, betas=(0.9, 0.95), eps=1e-6)
        loss_fn = (CRPSLoss(ds, 1.0, 2, 1.0) if kind == "crps" else
                   SCMLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), 1.0, tangent_warmup_kimg=1)).to(dev)
        tr = Trainer(net, opt, loss_fn, total_kimg=1, ema_halflife_kimg=1, ema_rampup_ratio=0.05, lr_rampup_kimg=0,
                     lr_min_factor=1.0, kimg_per_tick=1, checkpoint_ticks=None, device=dev)
        tr.global_batch_size = 2
        x = det_normal((2, 72, 64, 64), 35, "c").to(dev)
        t = (0.5 * det_normal((2, 69, 64, 64), 35, "t")).to(dev)
        delta, idx = torch.tensor([0.6, 0.6]), [0, 3]
        out = []
        for k in range(6):
            torch.manual_seed(100 + k)  # same noise draws in both runs
            out.append(float(tr.train_step(x, t, idx, delta, global_nimg=2 * (k + 1), steps=2)))
        return out

    eager, replay = run("0"), run("1")
    if os.environ.get("SWIFTK_TEST_NOISE_FLOOR"):  # how far do two EAGER runs drift apart (fp32 atomics through Adam)?
        print(kind, "eager2", [f"{v:.5f}" for v in run("0")])
    print(kind, "eager ", [f"{v:.5f}" for v in eager])
    print(kind, "replay", [f"{v:.5f}" for v in replay])
    assert all(math.isfinite(v) for v in eager + replay)
    for a, b in zip(eager, replay):
        assert b == pytest.approx(a, rel=2e-2, abs=2e-2)


def test_trainer_step_fused_kernel_vs_reference_golden(dev):
    """SURVEY section 8 row a18 with an oracle: swiftk_adamw_ema_step (gradient sanitising + AdamW + EMA in one pass) inside
    Trainer.train_step against the reference's own Trainer._backward_step on injected gradients with NaN / +-inf entries
    (tests/golden/trainer_tiny.npz), over warm-up, cosine and final learning rates; then checkpoint round trip of its state."""
    from test_oracle_golden import _run_trainer_on_fixture
    tr, worst = _run_trainer_on_fixture(dev)
    print(f"fused optimiser step vs reference golden: worst rel-L2 {worst:.3e}")
    assert tr._fused and worst < 3e-6
    sd = tr.optimizer.state_dict()
    assert len(sd["state"]) == 6 and float(sd["state"][0]["step"]) == 4.0
    assert sd["state"][0]["exp_avg"].shape == tr.optimizer.param_groups[0]["params"][0].shape


@pytest.mark.parametrize("dim,heads", [(1280, 16), (1536, 16)])  # head_dim 80 / 96: experiment/era5-swinv2-1.4-scm.yaml:29-36
def test_larger_variants_training_step_and_tangent_vs_oracle(dev, dim, heads):
    """the 468 M / 664 M widths (depth cut to 2) through the same training path as Swift-B -- QK-norm
    GEMM epilogue on 320- / 384-wide tiles, streamed attention, attention / QK-norm backward and the tangent kernels
    templated on head_dim; dim 1280 also has the odd MLP width int(8/3 * 1280) = 3413."""
    from oracle import loss as oloss
    from swift_amd.jvp_engine import SwinJvpEngine
    from swift_amd.models.precond import _process_auxiliary
    from swift_amd.training.loss import TrigFlowLoss
    from swift_amd.training.trainer import GradAllReduce
    from swift_amd.utils.detinit import det_normal
    net, onet, st = _build_pair(dev, 61, logvar=True, dim=dim, heads=heads)
    ds = _dataset(61)
    L = TrigFlowLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), sigma_data=1.0).to(dev)
    B = 2
    x, cond, z = det_normal((B, 69, 64, 64), 61, "x"), det_normal((B, 72, 64, 64), 61, "c"), det_normal((B, 69, 64, 64), 61, "z")
    tau, aux = torch.tensor([0.3, 4.0]).view(B, 1, 1, 1), torch.tensor([0.6, 0.6])
    ddp = GradAllReduce(net)
    ddp.zero_grad_flat()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = L(ddp, x.to(dev), condition=cond.to(dev), auxiliary=aux.to(dev), _tau=tau.to(dev), _z=z.to(dev))
    loss.backward()
    ref = oloss.trigflow_loss(onet, x, tau, z, L.w_var.cpu(), L.w_lat.cpu(), 1.0, condition=cond, auxiliary=aux, return_logvar=True)
    ref.backward()
    print(f"dim {dim}: trigflow loss {float(loss):.6f} vs oracle {float(ref):.6f}; worst grad cosine {_grad_report(net, st):.4f}")
    assert float(loss) == pytest.approx(float(ref), rel=1e-3)
    # the forward-mode tangent of the same network (sCM pre-training), fp32 and bf16 operands
    vx = det_normal((B, 69, 64, 64), 61, "vx")
    t, vt = torch.tensor([0.4, 1.3]), torch.tensor([0.35, 0.2])
    with torch.no_grad():
        Fref, dref = torch.func.jvp(lambda xx, tt: onet(xx, tt, cond, aux, jvp=True), (x, t), (vx, vt))
    auxd = _process_auxiliary(aux.to(dev), 1, B, dev)
    errs = {}
    for dt in (torch.float32, torch.bfloat16):
        dF = SwinJvpEngine(net.model, dt).jvp([x.to(dev), cond.to(dev)], vx.to(dev), t.to(dev), vt.to(dev), auxd)
        errs[dt] = rel_l2(dF.cpu(), dref)
    print(f"dim {dim}: network tangent vs oracle jvp: fp32 rel-L2 {errs[torch.float32]:.3e}, bf16 {errs[torch.bfloat16]:.3e}")
    assert errs[torch.float32] < 1e-4 and errs[torch.bfloat16] < 8e-2


def test_graph_replay_equals_eager(dev):
    """graphs.GraphCache: the first call of a signature runs eagerly, the second is captured into a HIP graph and replayed,
    later ones replay -- same loss and gradients each time (fp32 atomics in the loss mean and the LayerNorm / bias column
    sums are the only run-to-run noise: ~3e-6 of the loss between two eager runs), with fresh inputs picked up by the replays and updated weights by the persistent operand buffers."""
    from swift_amd.training.loss import TrigFlowLoss
    from swift_amd.training.trainer import GradAllReduce
    from swift_amd.utils.detinit import det_normal
    net, onet, st = _build_pair(dev, 77, logvar=True)
    ds = _dataset(77)
    L = TrigFlowLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), sigma_data=1.0).to(dev)
    B = 2
    ddp = GradAllReduce(net)
    tau, aux = torch.tensor([0.3, 4.0]).view(B, 1, 1, 1).to(dev), torch.tensor([0.6, 0.6]).to(dev)

    def run(tag):
        x, cond, z = (det_normal((B, 69, 64, 64), 77, f"x{tag}").to(dev), det_normal((B, 72, 64, 64), 77, f"c{tag}").to(dev),
                      det_normal((B, 69, 64, 64), 77, f"z{tag}").to(dev))
        ddp.zero_grad_flat()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = L(ddp, x, condition=cond, auxiliary=aux, _tau=tau, _z=z)
        loss.backward()
        return float(loss), ddp.flatten_grads().clone()

    eng = net.model._train_engine if hasattr(net.model, "_train_engine") else None
    l_a, g_a = run("a")           # eager
    eng = net.model._train_engine
    assert not eng.graphs._graphs
    l_a2, g_a2 = run("a")         # captured, then replayed
    assert len(eng.graphs._graphs) == 2  # forward + backward
    l_a3, g_a3 = run("a")         # replayed
    assert l_a2 == pytest.approx(l_a, rel=1e-5) and l_a3 == pytest.approx(l_a, rel=1e-5)
    assert rel_l2(g_a2.cpu(), g_a.cpu()) < 5e-5 and rel_l2(g_a3.cpu(), g_a.cpu()) < 5e-5
    l_b, g_b = run("b")           # other inputs through the same graphs ...
    assert abs(l_b - l_a) > 1e-4 and rel_l2(g_b.cpu(), g_a.cpu()) > 1e-2
    import os
    os.environ["SWIFTK_TRAIN_GRAPHS"] = "0"
    try:
        l_be, g_be = run("b")     # ... match the eager path on those inputs
    finally:
        del os.environ["SWIFTK_TRAIN_GRAPHS"]
    assert l_b == pytest.approx(l_be, rel=1e-5) and rel_l2(g_b.cpu(), g_be.cpu()) < 5e-5
    with torch.no_grad():         # a weight update is seen by the replays (operand copies are refreshed in place)
        for p in net.parameters():
            p.mul_(1.01)
    l_c, g_c = run("b")
    assert abs(l_c - l_b) > 1e-5 and len(eng.graphs._graphs) == 2


def test_graph_pools_survive_growing_rollouts_and_moved_gradients(dev, monkeypatch):
    """Two ways a captured launch sequence can go stale without an error:
    (1) the multistep schedule grows `steps` (1 -> 2 -> 4), so new resident slots are captured AFTER the old forward /
        backward sequences but replay BEFORE them: with one shared capture pool the old sequences' temporaries land in the
        new slots' saved activations -- every key that keeps outputs owns its pool now;
    (2) gradient buffers move (`zero_grad(set_to_none=True)`): the captured backward would add into freed memory -- the
        engine stamps parameter / gradient addresses and recaptures.
    Reference in both cases: the same calls with SWIFTK_TRAIN_GRAPHS=0."""
    from swift_amd.training.loss import CRPSLoss
    from swift_amd.utils.detinit import det_normal
    monkeypatch.delenv("SWIFTK_CRPS_KEEP", raising=False)
    B = 2
    ds = _dataset(41)
    target, cond = det_normal((B, 69, 64, 64), 41, "t").to(dev), det_normal((B, 72, 64, 64), 41, "c").to(dev)
    aux, idx = torch.tensor([0.6, 0.6]), [0, 4]
    lat = [[det_normal((B, 69, 64, 64), 41, f"l{e}{i}") for i in range(4)] for e in range(2)]

    def trajectory(graphs_on, set_to_none):
        monkeypatch.setenv("SWIFTK_TRAIN_GRAPHS", "1" if graphs_on else "0")
        net, _, _ = _build_pair(dev, 41)
        L = CRPSLoss(ds, sigma_data=1.0, ensemble_size=2, alpha=0.95).to(dev)
        out = []
        for steps in (1, 1, 1, 2, 2, 2, 4, 4, 4, 2, 1):     # every signature: eager, captured, replayed; then back down
            if set_to_none:
                net.zero_grad(set_to_none=True)
            else:
                for p in net.parameters():
                    if p.grad is not None:
                        p.grad.zero_()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = L(net, target, condition=cond, auxiliary=aux, idx=idx, steps=steps, _latents=[l[:steps] for l in lat])
            loss.backward()
            out.append((float(loss), torch.cat([p.grad.flatten() for p in net.parameters()]).cpu()))
        if graphs_on:
            cache = net.model._train_engine.graphs
            if set_to_none:  # gradient buffers move every iteration: every capture is dropped again, none may survive stale
                assert max(cache.generation.values()) > 1, "moved gradient buffers did not invalidate the captures"
            else:
                assert cache._graphs, "nothing was captured"
        return out

    ref = trajectory(False, False)
    for set_to_none in (False, True):
        got = trajectory(True, set_to_none)
        for k, ((l0, g0), (l1, g1)) in enumerate(zip(ref, got)):
            assert l1 == pytest.approx(l0, rel=1e-5), (set_to_none, k)
            assert rel_l2(g1, g0) < 1e-4, (set_to_none, k, rel_l2(g1, g0))


def test_repeat_screen_of_hand_synchronised_training_kernels(dev):
    """The paired-row GEMM epilogues, the split-output SwiGLU epilogue, the staged attention tangent kernel and the attention
    backward with its counted store wait: launched dozens of times on the same inputs, every output bit-equal to the first
    launch's (tools/repeat_screen.py, tools/attn_bwd_repeat.py; the forecast kernels' screen is tools/stress.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "repeat_screen.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "REPEAT SCREEN: CLEAN" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "attn_bwd_repeat.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "mismatching outputs: 0" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.timeout(600)
def test_crps_finetune_memory_rule_at_the_664m_variant(dev):
    """BASELINE configs[4]'s iteration at the reference's largest commented variant (dim 1536, 16 heads, depth 16;
    era5-swinv2-1.4-scm.yaml:29-36), local batch 8: all eight rollout steps resident would need ~420 GiB, so the planner
    (loss.py::_plan_once) must keep only what fits and recompute the rest -- like the reference's checkpoint_sequential --
    instead of running out of memory.  Run as the tool the bench line uses, in a fresh process (its own allocator)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.empty_cache()  # (the child plans around what this process still holds)
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "train_bench.py"), "--loss", "crps", "--dim", "1536", "--heads", "16",
                        "--depth", "16", "--iters", "2"], capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("{")))
    print(f"CRPS finetune at dim 1536 / depth 16: {rec['value']:.3f} s per iteration, {rec['kept_rollout_steps']} of 8 rollout steps resident, "
          f"peak {rec['peak_mem_gib']:.0f} GiB, {rec['roofline']['frac']:.3f} of the dense bf16 peak")
    assert 0 <= rec["kept_rollout_steps"] < 8         # not everything resident (that would need ~420 GiB): the rest is recomputed
    assert rec["peak_mem_gib"] < 0.9 * 288            # well inside the device
    assert rec["roofline"]["frac"] > 0.25 and rec["allreduce"]["iterations_recorded"] == 2
