"""-m gpu: the whole forecast path on the MI355X against (a) golden vectors produced by the actual
reference and (b) the CPU oracle run live on the same inputs.

Tolerances (relative L2):
  fp32 engine  1e-4  -- the bar BASELINE.json's north_star states ("within 1e-4 relative L2 of the reference")
  bf16 engine  calibrated, not guessed: the reference run under its own bf16 autocast sits 6.4e-2 (depth 2) / 1.9e-1
               (Swift-B, depth 12) from its fp32 output on these random-weight nets (logit scales up to 100 make the
               softmax sensitive to operand rounding); the bf16 engine must be no further from the fp32 reference than
               1.25 x that distance (fixtures store it as bf16_autocast_rel).  Reported, not the parity bar.
  bf16 engine vs the oracle's bf16-operand emulation (round 4; oracle/swinv2.py ``emulate_bf16``: the operands of every large
               Linear, of QK^T and of PV rounded to bf16, fp32 accumulation / norm / softmax / residual): BF16_EMU_TOL -- the
               distance that is left when both sides round the same quantities, i.e. what a defect would have to hide under.
  fp64 anchor  (round 4; tests/golden/swiftb_fp64.npz = the reference itself in fp64): where two fp32 implementations differ by
               more than 1e-4 after many chained evaluations, the engine must be no further from the fp64 truth than 1.5 x the
               reference's own fp32 run is.
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from swift_amd.utils.detinit import det_normal, swinv2_state

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4
BF16_EMU_TOL = 2e-2  # bf16 engine vs the bf16-operand emulation of the oracle (one network evaluation, depth 2)
# (depth 12: calibrated inside test_swiftb_full_step_vs_reference_golden on the distance between two admissible emulations)
ROLLOUT_BF16_EMU_TOL = 3e-2  # increments of a 4-step rollout (errors feed back through the state)
BF16_TOL = 1.0e-1  # depth-2 nets: 1.5 x the reference's own bf16-autocast-vs-fp32 distance (6.4e-2, tests/golden/swinv2_smallb.npz)

SMALLB = dict(img=(64, 64), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1056, heads=12, depth=2)
SWIFTB = dict(img=(128, 256), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1056, heads=12, depth=12)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def build(c, seed, dev, logvar=False):
    """(product net on the GPU, oracle net on the CPU) sharing one deterministic state dict."""
    from oracle.swinv2 import OracleNet, SwinCfg
    from swift_amd.models.precond import PassPrecond
    nv, nf = c["n_vars"], c["n_forc"]
    mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=list(c["window"]), shift_size=list(c["shift"]),
                patch_size=list(c["patch"]), depth=c["depth"], dim=c["dim"], heads=c["heads"], logvar=logvar)
    net = PassPrecond(mcfg, img_resolution=list(c["img"]), img_channels=nv, condition_channels=nv + nf, auxiliary_dim=1,
                      sigma_min=0, sigma_max=float("inf"), sigma_data=1.0)
    grid = (c["img"][0] // c["patch"][0], c["img"][1] // c["patch"][1])
    state = swinv2_state(grid=grid, in_channels=2 * nv + nf, out_channels=nv, patch_size=c["patch"], depth=c["depth"],
                         dim=c["dim"], heads=c["heads"], auxiliary_dim=1, logvar=logvar, seed=seed)
    net.load_state_dict(state, strict=True)
    net = net.to(dev).eval()
    ocfg = SwinCfg(img_resolution=c["img"], in_channels=2 * nv + nf, out_channels=nv, window_size=c["window"],
                   shift_size=c["shift"], patch_size=c["patch"], depth=c["depth"], dim=c["dim"], heads=c["heads"],
                   auxiliary_dim=1, logvar=logvar)
    return net, OracleNet(ocfg, state, img_channels=nv, condition_channels=nv + nf)


def test_forward_vs_reference_golden(dev):
    g = load_golden("swinv2_smallb")
    net, onet = build(SMALLB, int(g["seed"]), dev)
    x = det_normal((2, 141, 64, 64), int(g["seed"]), "x")
    t, aux = torch.from_numpy(g["t"]), torch.from_numpy(g["aux"])
    with torch.no_grad():
        y = net.model(x.to(dev), t.to(dev), auxiliary=aux.to(dev))
        yb = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = net.model(x.to(dev), t.to(dev), auxiliary=aux.to(dev))
        net.model.fp32_engine = "bf16x3"  # fp32 activations, every GEMM as three bf16 products of (hi, lo)-split operands
        y3 = net.model(x.to(dev), t.to(dev), auxiliary=aux.to(dev))
        net.model.fp32_engine = None
    e32, e16, e3 = rel_l2(y.cpu(), g["y_flash"]), rel_l2(yb.cpu(), g["y_flash"]), rel_l2(y3.cpu(), g["y_flash"])
    print(f"forward smallb: fp32 rel-L2 {e32:.3e}, bf16x3 rel-L2 {e3:.3e}, bf16 rel-L2 {e16:.3e}")
    assert y.dtype == torch.float32 and e32 < FP32_TOL
    assert y3.dtype == torch.float32 and e3 < FP32_TOL and not torch.equal(y3, y)  # (a different engine did run)
    assert e16 < 1.25 * float(g["bf16_autocast_rel"])
    # the tight yardstick for the bf16 engine: the oracle with the same operand roundings
    from oracle.swinv2 import swinv2_forward
    with torch.no_grad():
        yemu = swinv2_forward(onet.cfg, onet.p, x, t, auxiliary=aux, emulate_bf16=True)
    eemu, demu = rel_l2(yb.cpu(), yemu), rel_l2(yemu, g["y_flash"])
    print(f"forward smallb: bf16 engine vs bf16-emulating oracle {eemu:.3e} (emulation vs fp32 reference {demu:.3e})")
    assert eemu < BF16_EMU_TOL and eemu < 0.5 * e16
    # this small grid (24 output tiles for wo / w2) takes the split-K form of those GEMMs; the large-batch form -- one launch,
    # bf16 y, what the benchmark runs -- on the same inputs, and the fp32-stream form of the residual (tuning keys 14, 12)
    from swift_amd import _lib
    L = _lib.lib()
    # (key 29 = 0: the generic two-slab norm kernel behind the split-K instead of the packed one that adds the halves itself)
    for key, val in ((14, 0), (12, 1), (12, 0), (29, 0)):
        old = L.swiftk_get_tuning(key)
        L.swiftk_set_tuning(key, val)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            yv = net.model(x.to(dev), t.to(dev), auxiliary=aux.to(dev))
        L.swiftk_set_tuning(key, old)
        ev = rel_l2(yv.cpu(), yemu)
        print(f"  tuning {key}:{val}: bf16 engine vs bf16-emulating oracle {ev:.3e}; vs the default form {rel_l2(yv.cpu(), yb.cpu()):.3e}")
        assert ev < BF16_EMU_TOL and not torch.equal(yv, yb)


@pytest.mark.parametrize("units", [3, 4, 6])
def test_forward_small_batch_tail_split(dev, units):
    """Round 6: at 3 / 4 / 6 units per step wo / w2 leave the persistent walk a last round that is at most half full; the engine runs
    that round's tiles as two k-halves (swiftk_gemm_tail_split_bf16 + swiftk_modnorm_residual_pair_halves_bf16, tuning key 29 bit 0).
    Same network, same inputs, with and without: equal up to the halves' extra bf16 rounding; and a step is repeatable."""
    from swift_amd import _lib
    L = _lib.lib()
    net, _ = build(dict(SWIFTB, depth=2), 5, dev)
    x = det_normal((units, 141, 128, 256), 9, "x").to(dev)
    t = torch.linspace(0.2, 1.4, units).to(dev)
    aux = torch.full((units, 1), 0.6).to(dev)

    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return net.model(x, t, auxiliary=aux).float()

    y_tail, y_again = run(), run()
    old = L.swiftk_get_tuning(29)
    L.swiftk_set_tuning(29, old & ~1)
    try:
        y_plain = run()
    finally:
        L.swiftk_set_tuning(29, old)
    # with the GEMM's ping-pong loop switched off (tuning key 20) the k-half walk does not exist: the engine must fall back to whole
    # tiles by itself -- and that loop is bit-equal to the ping-pong one, so the result is the whole-tile result
    L.swiftk_set_tuning(20, 0)
    try:
        y_fallback = run()
    finally:
        L.swiftk_set_tuning(20, 1)
    assert torch.equal(y_fallback, y_plain)
    e = rel_l2(y_tail.cpu(), y_plain.cpu())
    print(f"forward, {units} units, depth 2: last round as k-halves vs whole tiles rel-L2 {e:.3e}")
    assert torch.isfinite(y_tail).all() and torch.equal(y_tail, y_again)
    assert e < 1e-2 and not torch.equal(y_tail, y_plain)  # (two bf16 forms of the same network: 5e-3 apart at depth 2; the other path did run)


@pytest.mark.parametrize("name,c", [
    # reference experiment/era5-swinv2-5.6-scm.yaml: 1x1 patches on the 32x64 grid (head width 69: not a multiple of 4)
    ("5.6deg-1x1-patches", dict(img=(32, 64), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(1, 1), dim=1056,
                                heads=12, depth=2)),
    # the larger variants in experiment/era5-swinv2-1.4-scm.yaml's comments: head_dim 80 (1280/16) and 96 (1536/16)
    ("head_dim-80", dict(img=(64, 64), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=320, heads=4,
                         depth=2)),
    ("head_dim-96", dict(img=(64, 64), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=384, heads=4,
                         depth=2)),
    # ... and at their real widths (16 heads; depth cut to 2): bf16 runs the QK-norm epilogue on 320- / 384-wide GEMM tiles,
    # the window-tiled q/k/v store and the streamed attention kernel templated on head_dim; dim 1280 has an odd MLP width
    ("468M-width-dim1280-hd80", dict(img=(64, 64), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1280,
                                     heads=16, depth=2)),
    ("664M-width-dim1536-hd96", dict(img=(64, 64), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1536,
                                     heads=16, depth=2)),
])
def test_forward_other_swift_variants_vs_oracle(dev, name, c):
    net, onet = build(c, 12, dev)
    x, cond = det_normal((2, 69, *c["img"]), 12, "x"), det_normal((2, 72, *c["img"]), 12, "cond")
    t = torch.tensor([0.4, 1.3])
    with torch.no_grad():
        y = net(x.to(dev), t.to(dev), cond.to(dev), 0.6)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = net(x.to(dev), t.to(dev), cond.to(dev), 0.6)
        yo = onet(x, t, cond, 0.6)
        net.model.fp32_engine = "bf16x3"  # (1x1 patches: 141 input columns rounded up to 144; dim 1280: odd MLP width)
        y3 = net(x.to(dev), t.to(dev), cond.to(dev), 0.6)
        net.model.fp32_engine = None
    e32, e16, e3 = rel_l2(y.cpu(), yo), rel_l2(yb.cpu(), yo), rel_l2(y3.cpu(), yo)
    print(f"{name}: fp32 rel-L2 {e32:.3e}, bf16x3 rel-L2 {e3:.3e}, bf16 rel-L2 {e16:.3e}")
    assert e32 < FP32_TOL and e16 < BF16_TOL and e3 < FP32_TOL
    # the split engine's to_qkv is the ADAPTIVE form at every head_dim the QK-norm epilogue serves (80 / 88 / 96): split
    # products, the head pairs with a large logit scale recomputed on the exact-fp32 kernel (mask bit 6 kept, bit 0 clear) --
    # and these nets do have such pairs (scales up to ln 100)
    eng3 = net.model._engines["bf16x3"]
    mask = int(eng3.model.x3_exact)
    hot = [int(eng3.model.layers_host[i].qk_exact_pairs) for i in range(c["depth"])]
    print(f"  split engine: x3_exact mask {mask}, hot head pairs per layer {hot}")
    assert (mask & 64) and not (mask & 1) and any(hot)


def test_forward_batch16_vs_oracle(dev):
    """From 16 samples on the 2 x depth modulation Linears run as ONE fp32 MFMA GEMM instead of the small-batch VALU kernel
    (csrc/forward.hip); 16 is also several items per workgroup for the fused to_qkv + attention kernel.  Per-sample t and
    auxiliary values differ, so a row mix-up in the modulation matrix would show."""
    net, onet = build(SMALLB, 21, dev)
    B = 16
    x, cond = det_normal((B, 69, 64, 64), 21, "x"), det_normal((B, 72, 64, 64), 21, "cond")
    t = torch.linspace(0.1, 1.5, B)
    aux = torch.tensor([0.6, 1.2, 2.4, 0.6] * 4)
    with torch.no_grad():
        y = net(x.to(dev), t.to(dev), cond.to(dev), aux.to(dev))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = net(x.to(dev), t.to(dev), cond.to(dev), aux.to(dev))
        yo = onet(x, t, cond, aux)
    e32, e16 = rel_l2(y.cpu(), yo), rel_l2(yb.cpu(), yo)
    print(f"batch 16: fp32 rel-L2 {e32:.3e}, bf16 rel-L2 {e16:.3e}")
    assert e32 < FP32_TOL and e16 < BF16_TOL
    for b in (0, 7, 15):  # per sample, not only in aggregate
        assert rel_l2(y[b].cpu(), yo[b]) < FP32_TOL


def test_forward_logvar_and_split_sources(dev):
    net, onet = build(SMALLB, 8, dev, logvar=True)
    x, cond = det_normal((2, 69, 64, 64), 8, "x"), det_normal((2, 72, 64, 64), 8, "cond")
    t = torch.tensor([0.4, 1.3])
    with torch.no_grad():
        y, lv = net(x.to(dev), t.to(dev), cond.to(dev), 0.6, return_logvar=True)
        yo, lvo = onet(x, t, cond, 0.6, return_logvar=True)
        y2 = net(x.to(dev), t.to(dev), (cond[:, :69].to(dev), cond[:, 69:].to(dev)), 0.6)
    assert rel_l2(y.cpu(), yo) < FP32_TOL and rel_l2(lv.cpu(), lvo) < FP32_TOL
    assert torch.equal(y, y2)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)])
def test_samplers_vs_oracle(dev, dtype, tol):
    from oracle import sampler as osamp
    from swift_amd.generating.factory import sampler_factory
    net, onet = build(SMALLB, 9, dev)
    B = 2
    cond, lat = det_normal((B, 72, 64, 64), 9, "cond"), det_normal((B, 69, 64, 64), 9, "lat")
    ren = [det_normal((B, 69, 64, 64), 9, f"ren{i}") for i in range(3)]
    kw = dict(sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    for n in (1, 2, 3):
        it = iter(ren)
        smp = sampler_factory("scm", net, denoise_dtype=dtype, num_steps=n, randn_like=lambda like: next(it).to(like), **kw)
        y = smp(cond.to(dev), latents=lat.to(dev))
        ref = osamp.scm_solver(onet, lat, cond, renoise=ren, num_steps=n, **kw)
        e = rel_l2(y.cpu(), ref)
        print(f"scm num_steps={n} {dtype}: rel-L2 {e:.3e}")
        assert e < tol * (1 if n == 1 else 3)
    smp = sampler_factory("2s", net, denoise_dtype=dtype, num_steps=3, **kw)
    y = smp(cond.to(dev), latents=lat.to(dev))
    ref = osamp.dpm_solver_2s(onet, lat, cond, num_steps=3, **kw)
    e = rel_l2(y.cpu(), ref)
    print(f"dpm_solver_2s num_steps=3 {dtype}: rel-L2 {e:.3e}")
    assert e < tol * 3


def test_dpm_solver_and_validation_rollout_vs_oracle(dev):
    """dpm_solver (diffusion.py:289-353) and RMSE_rollout (training/validate.py:23-127) on the fp32 engine against their
    CPU restatements (which tests/test_oracle_golden.py pins to the reference's own functions)."""
    from oracle import rollout as oroll
    from oracle import sampler as osamp
    from oracle import validate as oval
    from swift_amd.data.era5 import SyntheticERA5RollOutDataset
    from swift_amd.generating.factory import sampler_factory
    from swift_amd.training.validate import RMSE_rollout
    net, onet = build(SMALLB, 9, dev)
    B = 2
    cond, lat = det_normal((B, 72, 64, 64), 9, "cond"), det_normal((B, 69, 64, 64), 9, "lat")
    kw = dict(sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    for pp in (True, False):
        y = sampler_factory("dpm", net, num_steps=4, use_pp=pp, **kw)(cond.to(dev), latents=lat.to(dev))
        e = rel_l2(y.cpu(), osamp.dpm_solver(onet, lat, cond, num_steps=4, use_pp=pp, **kw))
        print(f"dpm_solver num_steps=4 use_pp={pp}: rel-L2 {e:.3e}")
        assert e < 3e-4
    names = [f"v{i}" for i in range(69)]
    ds = SyntheticERA5RollOutDataset(8, names, ["f0", "f1", "f2"], img_resolution=(64, 64), length=40, seed=9, random_stats=True)
    items = [ds[0], ds[5]]
    X0, TS, idx = torch.stack([i[0] for i in items]), torch.stack([i[1] for i in items]), [0, 5]
    assert TS.shape == (2, 3, 69, 64, 64) and torch.equal(TS[0, 0], ds._fields(1, "state", 69)) \
        and torch.equal(TS[0, 2], ds._fields(8, "state", 69))
    g = torch.Generator(device=dev).manual_seed(5)
    lats = [torch.randn((B, 69, 64, 64), generator=g, device=dev) for _ in range(8)]
    g = torch.Generator(device=dev).manual_seed(5)
    agg, sep = RMSE_rollout(sampler_factory("dpm", net, num_steps=2, **kw), iter([(X0, TS, idx)]), ds, 8, dev, rng=g,
                            num_batches=1)
    stats = oroll.Stats(ds.x_means, ds.x_stds, {6: ds.t_stds[6]}, n_vars=69, n_forc=3)
    it = iter(lats)
    ragg, rsep = oval.rmse_rollout(lambda c: osamp.dpm_solver(onet, next(it).cpu(), c, num_steps=2, **kw), stats, X0, TS,
                                   lambda i: torch.stack([ds.get_forcings(j + i) for j in idx], 0), ds.get_lat_lon()[0], 8)
    print(f"validation rollout (8 steps): aggregate RMSE {agg:.6f} vs oracle {ragg:.6f}")
    assert agg == pytest.approx(ragg, rel=2e-4)
    np.testing.assert_allclose(sep, rsep, rtol=1e-3)


@pytest.mark.parametrize("cfg,B", [(SMALLB, 2), (dict(SWIFTB, depth=2), 4)], ids=["smallb-2-units", "full-grid-4-units"])
def test_hip_graph_step_equals_eager(dev, cfg, B):
    """RolloutEngine.capture_step: replaying the recorded step advances the state exactly as the eager launches do.  (Four units on
    the full grid: wo / w2 take the last-round-as-k-halves path, whose walk description travels by value in the launches.)"""
    from swift_amd import ops
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import RolloutEngine
    net, _ = build(cfg, 9, dev)
    H, W = cfg["img"]
    ds = SyntheticERA5Dataset([f"v{i}" for i in range(69)], ["f0", "f1", "f2"], img_resolution=(H, W), length=16, seed=9)
    eng = RolloutEngine(net, ds, interval=6, solver="scm", denoise_dtype=torch.bfloat16)
    X0 = det_normal((B, 69, H, W), 9, "X0").to(dev)
    forc = det_normal((B, 3, H, W), 9, "f").to(dev)
    zs = [det_normal((B, 69, H, W), 9, f"z{i}").to(dev) for i in range(3)]
    mx, sx, st = eng.stats(dev)
    Xe, phys_e = X0.clone(), torch.empty_like(X0)
    for z in zs:
        ops.rollout_update(Xe, eng.sampler((Xe, forc), latents=z), mx, sx, st, phys=phys_e)
    Xg, zbuf, phys_g = X0.clone(), torch.empty_like(X0), torch.empty_like(X0)
    graph = eng.capture_step(Xg, forc, zbuf, phys_g)
    assert torch.equal(Xg, X0)
    for z in zs:
        zbuf.copy_(z)
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(Xg, Xe) and torch.equal(phys_g, phys_e)


def test_sampler_draws_like_reference(dev):
    """generating/factory.py:52-56: latents = torch.randn(shape, generator=g, device=X.device)."""
    from swift_amd.generating.factory import sampler_factory
    net, _ = build(SMALLB, 9, dev)
    cond = det_normal((1, 72, 64, 64), 9, "cond").to(dev)
    smp = sampler_factory("scm", net, num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    g = torch.Generator(device=dev).manual_seed(5)
    y1 = smp(cond, generator=g)
    g = torch.Generator(device=dev).manual_seed(5)
    lat = torch.randn((1, 69, 64, 64), generator=g, device=dev)
    assert torch.equal(y1, smp(cond, latents=lat))


def test_rollout_vs_oracle(dev):
    from oracle import rollout as oroll
    from oracle import sampler as osamp
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import RolloutEngine
    net, onet = build(SMALLB, 10, dev)
    ds = SyntheticERA5Dataset([f"v{i}" for i in range(69)], ["f0", "f1", "f2"], img_resolution=(64, 64), length=32,
                              seed=10, random_stats=True)
    steps, B, idx = 3, 2, [0, 5]
    X0 = det_normal((B, 69, 64, 64), 10, "X0")
    lats = [det_normal((B, 69, 64, 64), 10, f"lat{i}") for i in range(steps)]
    eng = RolloutEngine(net, ds, interval=6)
    forc = eng.stage_forcings(idx, steps, dev)
    traj = eng.run(X0.to(dev), forc, steps, latents=lambda i: lats[i].to(dev))
    stats = oroll.Stats(ds.x_means, ds.x_stds, {6: ds.t_stds[6]}, n_vars=69, n_forc=3)
    it = iter(lats)
    osampler = lambda c: osamp.scm_solver(onet, next(it), c, 0.6, num_steps=1, sigma_min=0.02, sigma_max=200.0)
    ref = oroll.rollout(osampler, stats, X0, lambda i: torch.stack([ds.get_forcings(j + i) for j in idx], 0), steps)
    assert traj.shape == ref.shape
    e = rel_l2(traj.cpu(), ref)
    print(f"3-step rollout fp32: rel-L2 {e:.3e}")
    assert e < FP32_TOL
    # sharding/batching independence: unit-keyed noise gives the same trajectory alone or in a batch
    t_all = eng.run(X0.to(dev), forc, 2, seeds=[11, 22])
    t_one = eng.run(X0[1:].to(dev), forc[:, 1:], 2, seeds=[22])
    assert torch.equal(t_all[1], t_one[0])
    # ... also with the multi-step consistency sampler, whose re-noising draws (diffusion.py:452-455) come from the same
    # counter-based stream (a global torch generator would make them depend on what ran before)
    eng3 = RolloutEngine(net, ds, interval=6, num_steps=3)
    torch.manual_seed(1)
    t_all = eng3.run(X0.to(dev), forc, 2, seeds=[11, 22])
    torch.manual_seed(2)
    t_one = eng3.run(X0[1:].to(dev), forc[:, 1:], 2, seeds=[22])
    assert torch.equal(t_all[1], t_one[0]) and not torch.equal(t_all[0], t_all[1])


def test_rollout_non_residual_dataset_vs_oracle(dev):
    """generate.py:132-136 / validate.py:112-116: a dataset whose targets are states (residual=False) -- the network output is
    the next standardised state and the trajectory holds ``unstandardize_x`` of it (bit-exact: one multiply, one add)."""
    from oracle import rollout as oroll
    from oracle import sampler as osamp
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import RolloutEngine
    net, onet = build(SMALLB, 10, dev)
    ds = SyntheticERA5Dataset([f"v{i}" for i in range(69)], ["f0", "f1", "f2"], img_resolution=(64, 64), length=32,
                              seed=10, random_stats=True, residual=False)
    steps, B, idx = 3, 2, [0, 5]
    X0 = det_normal((B, 69, 64, 64), 10, "X0")
    lats = [det_normal((B, 69, 64, 64), 10, f"lat{i}") for i in range(steps)]
    eng = RolloutEngine(net, ds, interval=6)
    forc = eng.stage_forcings(idx, steps, dev)
    traj = eng.run(X0.to(dev), forc, steps, latents=lambda i: lats[i].to(dev))
    stats = oroll.Stats(ds.x_means, ds.x_stds, {6: ds.x_stds[:69]}, n_vars=69, n_forc=3)
    it = iter(lats)
    osampler = lambda c: osamp.scm_solver(onet, next(it), c, 0.6, num_steps=1, sigma_min=0.02, sigma_max=200.0)
    ref = oroll.rollout(osampler, stats, X0, lambda i: torch.stack([ds.get_forcings(j + i) for j in idx], 0), steps, residual=False)
    e = rel_l2(traj.cpu(), ref)
    print(f"3-step non-residual rollout fp32: rel-L2 {e:.3e}")
    assert traj.shape == ref.shape and e < FP32_TOL
    # the update itself, bit for bit: phys = y * sx + mx (two roundings), state = y
    from swift_amd import ops
    mx, sx, st = eng.stats(dev)
    assert st is None
    y, x, phys = det_normal((B, 69, 64, 64), 10, "y").to(dev), det_normal((B, 69, 64, 64), 10, "x").to(dev), torch.empty(B, 69, 64, 64, device=dev)
    ops.rollout_update(x, y, mx, sx, None, phys=phys)
    assert torch.equal(x, y) and torch.equal(phys, y * sx.view(1, -1, 1, 1) + mx.view(1, -1, 1, 1))


def test_swiftb_full_step_vs_reference_golden(dev):
    """BASELINE config 1 on the GPU: Swift-B, 1 member x 1 IC x 1 step, fp32 engine vs the reference's output."""
    from swift_amd.generating.factory import sampler_factory
    g = load_golden("swiftb_step")
    seed = int(g["seed"])
    net, onet = build(SWIFTB, seed, dev)
    cond, lat = det_normal((1, 72, 128, 256), seed, "cond"), det_normal((1, 69, 128, 256), seed, "lat")
    kw = dict(num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    y = sampler_factory("scm", net, **kw)(cond.to(dev), latents=lat.to(dev))
    e32 = rel_l2(y[0, ::4, ::8, ::8].cpu(), g["y_sub"])
    yb = sampler_factory("scm", net, denoise_dtype=torch.bfloat16, **kw)(cond.to(dev), latents=lat.to(dev))
    e16 = rel_l2(yb[0, ::4, ::8, ::8].cpu(), g["y_sub"])
    net.model.fp32_engine = "bf16x3"
    y3 = sampler_factory("scm", net, **kw)(cond.to(dev), latents=lat.to(dev))
    net.model.fp32_engine = None
    e3 = rel_l2(y3[0, ::4, ::8, ::8].cpu(), g["y_sub"])
    print(f"Swift-B scm step vs reference: fp32 rel-L2 {e32:.3e}, bf16x3 rel-L2 {e3:.3e}, bf16 rel-L2 {e16:.3e}")
    assert e32 < FP32_TOL and e3 < FP32_TOL  # the north star's 1e-4, by either fp32-grade engine
    assert float(y.double().norm()) == pytest.approx(float(g["stats"][3]), rel=1e-4)
    assert e16 < 1.25 * float(g["bf16_autocast_rel"])  # reference's own bf16 path: 1.9e-1 at depth 12
    # fp64 anchor: the reference run in fp64 on the same inputs (tools/make_golden.py::fx_swiftb_fp64)
    g64 = load_golden("swiftb_fp64")
    d32, d3, dref = (rel_l2(y[0, ::4, ::8, ::8].cpu(), g64["y1_sub"]), rel_l2(y3[0, ::4, ::8, ::8].cpu(), g64["y1_sub"]),
                     float(g64["ref_fp32_vs_fp64_step"]))
    print(f"Swift-B scm step vs the reference in fp64: exact engine {d32:.3e}, bf16x3 {d3:.3e}, the reference's own fp32 run {dref:.3e}")
    assert d32 < max(1.5 * dref, FP32_TOL) and d3 < FP32_TOL
    # bf16 engine vs the oracle with the same operand roundings (CPU, ~10 s)
    # Twelve layers with logit scales up to 100 amplify ANY difference in rounding: two emulations that differ only in an
    # admissible choice -- the softmax offset, row maximum vs zero (softmax is shift-invariant) -- end 4.3e-2 apart, a third
    # of either one's distance from fp32.  That distance, computed here, is the noise floor of a bf16 implementation of this
    # network; the engine must sit within 1.5 x of it from the emulation (measured: 4.3e-2 against a floor of 4.3e-2).
    from oracle.sampler import scm_solver
    sub = lambda v: v[0, ::4, ::8, ::8]
    onet.emulate_bf16 = True
    yemu = scm_solver(onet, lat, cond, 0.6, num_steps=1, sigma_min=0.02, sigma_max=200.0)
    onet.emulate_bf16 = "offset0"
    yemu0 = scm_solver(onet, lat, cond, 0.6, num_steps=1, sigma_min=0.02, sigma_max=200.0)
    onet.emulate_bf16 = False
    floor = rel_l2(sub(yemu0), sub(yemu))
    eemu, eemu0, demu = rel_l2(sub(yb).cpu(), sub(yemu)), rel_l2(sub(yb).cpu(), sub(yemu0)), rel_l2(sub(yemu), g["y_sub"])
    print(f"Swift-B scm step: bf16 engine vs bf16-emulating oracle {eemu:.3e} (row-max offset) / {eemu0:.3e} (offset 0 where the "
          f"logit bound allows, as the kernel); the two emulations from each other {floor:.3e}; emulation vs fp32 reference "
          f"{demu:.3e}; engine vs fp32 reference {e16:.3e}")
    assert min(eemu, eemu0) < 1.5 * floor and max(eemu, eemu0) < 2.0 * floor and floor < 0.5 * demu


def test_swiftb_dpm_2s_vs_reference_golden(dev):
    """BASELINE configs[2] at full size: dpm_solver_2s, num_steps 20 = 39 chained Swift-B evaluations of one sample, fp32 engine
    vs the reference's output (tests/golden/swiftb_long.npz).  39 evaluations compound the per-evaluation distance (~1e-5)."""
    from swift_amd.generating.factory import sampler_factory
    g = load_golden("swiftb_long")
    seed = int(g["seed"])
    net, _ = build(SWIFTB, seed, dev)
    cond, lat = det_normal((1, 72, 128, 256), seed, "cond"), det_normal((1, 69, 128, 256), seed, "lat")
    y = sampler_factory("2s", net, num_steps=20, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)(cond.to(dev), latents=lat.to(dev))
    e = rel_l2(y[0, ::4, ::8, ::8].cpu(), g["y2s_sub"])
    print(f"Swift-B dpm_solver_2s (39 evaluations) vs reference: fp32 rel-L2 {e:.3e}")
    # Two fp32 implementations of 39 chained evaluations: their distance is bounded by the north star's 1e-4 OR explained by
    # the fp64 truth (the reference itself in fp64, tests/golden/swiftb_fp64.npz) -- the engine must be no further from it
    # than 1.5 x the reference's own fp32 run is.  Never both loose.
    g64 = load_golden("swiftb_fp64")
    d_eng, d_ref = rel_l2(y[0, ::4, ::8, ::8].cpu(), g64["y2s_sub"]), float(g64["ref_fp32_vs_fp64_2s"])
    print(f"  vs the reference in fp64: engine {d_eng:.3e}, the reference's own fp32 run {d_ref:.3e}")
    assert e < FP32_TOL or d_eng < 1.5 * d_ref
    assert e < 3 * FP32_TOL
    assert float(y.double().norm()) == pytest.approx(float(g["y2s_norm"]), rel=3e-4)
    # the bf16 engine (the throughput configuration; to_qkv + attention fused, q/k/v never rounded through HBM) on the same 39
    # evaluations, bounded by what the reference's OWN bf16-autocast run of this sampler is away from its fp32 run
    # (tests/golden/swiftb_2s_bf16.npz: 9.3e-2 on the same sub-sample)
    y16 = sampler_factory("2s", net, denoise_dtype=torch.bfloat16, num_steps=20, sigma_min=0.02, sigma_max=200.0,
                          auxiliary=0.6)(cond.to(dev), latents=lat.to(dev))
    e16 = rel_l2(y16[0, ::4, ::8, ::8].cpu(), g["y2s_sub"])
    yard = float(load_golden("swiftb_2s_bf16")["bf16_autocast_rel_sub"])
    print(f"Swift-B dpm_solver_2s, bf16 engine vs reference fp32: rel-L2 {e16:.3e} (reference's own bf16 path: {yard:.3e})")
    assert torch.isfinite(y16).all() and e16 < 1.25 * yard


def test_swiftb_rollout_60_steps_vs_reference_golden(dev):
    """BASELINE configs[3], one unit at full length: 60 autoregressive six-hour steps of Swift-B through RolloutEngine.run.
    fp32 engine vs the reference's trajectory (generate.py:97-131 driven by hand, tools/make_golden.py::fx_swiftb_long) at every
    lead step -- the north star's "within 1e-4 relative L2 ... on the 60-step rollout" -- and the bf16 engine's drift from
    it, which is reported with a documented bound (bf16 is the throughput configuration, not the parity one)."""
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import RolloutEngine
    g = load_golden("swiftb_long")
    seed, steps = int(g["seed"]), 60
    net, _ = build(SWIFTB, seed, dev)
    bank = det_normal((steps + 2, 3, 128, 256), seed, "forc", std=1.5, mean=0.5)

    class DS(SyntheticERA5Dataset):
        def get_forcings(self, idx):
            return bank[int(idx)].clone()

    ds = DS([f"v{i}" for i in range(69)], ["f0", "f1", "f2"], img_resolution=(128, 256), length=steps + 8, seed=seed)
    ds.x_means, ds.x_stds = g["x_mean"], g["x_std"]
    ds.t_stds = {6: g["t_std6"]}
    ds.t_means = {6: np.zeros_like(g["t_std6"])}
    X0 = det_normal((1, 69, 128, 256), seed, "X0")
    ref = torch.from_numpy(g["traj_sub"])
    errs = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        eng = RolloutEngine(net, ds, interval=6, denoise_dtype=dt)
        forc = eng.stage_forcings([0], steps, dev)
        traj = eng.run(X0.to(dev), forc, steps, latents=lambda i: det_normal((1, 69, 128, 256), seed, f"lat{i}").to(dev))
        sub = traj[0, :, ::4, ::8, ::8].cpu()
        errs[name] = [rel_l2(sub[i], ref[i]) for i in range(steps + 1)]
        if name == "fp32":
            nrm = [float(traj[0, i].double().norm()) for i in (1, 30, 60)]
            assert nrm == pytest.approx([float(g["traj_norm"][i]) for i in (1, 30, 60)], rel=1e-4)
    e32, e16 = errs["fp32"], errs["bf16"]
    print("60-step Swift-B rollout vs reference, rel-L2 of the physical state at lead steps 1 / 10 / 30 / 60: "
          f"fp32 {e32[1]:.2e} / {e32[10]:.2e} / {e32[30]:.2e} / {e32[60]:.2e}; bf16 {e16[1]:.2e} / {e16[10]:.2e} / {e16[30]:.2e} / {e16[60]:.2e}")
    assert e32[0] < 1e-6 and max(e32) < FP32_TOL
    # bf16 drift bound: the physical state is dominated by its mean (|x_mean| ~ 2 sigma), which damps the relative distance
    assert max(e16) < 5e-2


def test_bf16_rollout_vs_bf16_emulating_oracle(dev):
    """The bf16 engine (the benchmarked configuration) against the oracle with the same operand roundings, per lead step of an
    autoregressive rollout: depth-2 Swift-B width, 4 steps, physical AND standardised state -- the bound a defect in the bf16
    kernels would have to hide under (the fp32 reference sits 4-6e-2 away from either; see the module docstring)."""
    from oracle import rollout as oroll
    from oracle import sampler as osamp
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import RolloutEngine
    net, onet = build(SMALLB, 10, dev)
    ds = SyntheticERA5Dataset([f"v{i}" for i in range(69)], ["f0", "f1", "f2"], img_resolution=(64, 64), length=32,
                              seed=10, random_stats=True)
    steps, B, idx = 4, 2, [0, 5]
    X0 = det_normal((B, 69, 64, 64), 10, "X0")
    lats = [det_normal((B, 69, 64, 64), 10, f"lat{i}") for i in range(steps)]
    eng = RolloutEngine(net, ds, interval=6, denoise_dtype=torch.bfloat16)
    forc = eng.stage_forcings(idx, steps, dev)
    traj = eng.run(X0.to(dev), forc, steps, latents=lambda i: lats[i].to(dev)).cpu()
    stats = oroll.Stats(ds.x_means, ds.x_stds, {6: ds.t_stds[6]}, n_vars=69, n_forc=3)
    fget = lambda i: torch.stack([ds.get_forcings(j + i) for j in idx], 0)
    refs = {}
    for emu in (True, False):
        onet.emulate_bf16 = emu
        it = iter(lats)
        refs[emu] = oroll.rollout(lambda c: osamp.scm_solver(onet, next(it), c, 0.6, num_steps=1, sigma_min=0.02, sigma_max=200.0),
                                  stats, X0, fget, steps)
    onet.emulate_bf16 = False
    # per lead step, on the residual of the step (state minus the initial state) so that the large mean does not hide it
    base = refs[True][:, :1]
    for i in range(1, steps + 1):
        e_emu = rel_l2(traj[:, i] - base[:, 0], refs[True][:, i] - base[:, 0])
        e_f32 = rel_l2(traj[:, i] - base[:, 0], refs[False][:, i] - base[:, 0])
        print(f"lead step {i}: bf16 engine vs bf16-emulating oracle {e_emu:.3e}; vs fp32 oracle {e_f32:.3e} (increments)")
        assert e_emu < ROLLOUT_BF16_EMU_TOL
    assert rel_l2(traj, refs[True]) < 2e-3  # the physical state itself


@pytest.mark.parametrize("name,c", [
    ("Swift-B", SWIFTB),
    # the reference's larger variants at their real widths (depth cut to 2, a quarter of the grid): the fused kernel's head_dim 80 /
    # 96 geometries, the 320- / 384-wide GEMM tiles and the odd MLP width of dim 1280 inside the model's own call sequence
    ("468M-width-dim1280-hd80", dict(img=(64, 128), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1280,
                                     heads=16, depth=2)),
    ("664M-width-dim1536-hd96", dict(img=(64, 128), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1536,
                                     heads=16, depth=2)),
])
def test_swiftb_bf16_every_layer_teacher_forced_vs_emulating_oracle(dev, name, c):
    """Full-depth bf16 parity, layer by layer.  The end-to-end bound of test_swiftb_full_step_vs_reference_golden is calibrated on
    a noise floor of 4.3e-2 (twelve layers amplify any admissible rounding difference), which would hide a kernel defect of that
    size.  Here every one of Swift-B's 24 branches is judged on its own: the bf16-emulating oracle runs the whole network once
    (CPU), its residual stream in front of each branch is handed to the ENGINE's kernels for that branch -- the calls
    csrc/forward.hip makes: swiftk_qkv_attention_fused, swiftk_gemm (wo / w1 + SwiGLU / w2), swiftk_modnorm_residual_pair on the
    (bf16, 8-bit) pair -- and the branch output (what the branch adds to the stream) must agree with the oracle's to 1.5e-2.
    Nothing feeds back, so a branch's figure is that branch's (swinv2.py:186-212)."""
    import torch.nn.functional as F
    from swift_amd import ops
    seed = 21
    net, onet = build(c, seed, dev)
    m = net.model
    grid = (c["img"][0] // 2, c["img"][1] // 2)
    d, heads, n, hd = m.dim, m.heads, grid[0] * grid[1], m.dim // m.heads
    kd = ops.k_pad(torch.bfloat16, d)
    x = det_normal((1, 69, *c["img"]), seed, "x")
    cond = det_normal((1, 72, *c["img"]), seed, "cond")
    taps = {}
    onet(x, torch.tensor([1.1]), condition=cond, auxiliary=0.6, taps=taps, emulate_bf16="offset0")
    lat = taps["lat"].to(dev)
    worst = {"attention": 0.0, "feed-forward": 0.0}
    for i, (att, ff) in enumerate(m.transformer.layers):
        x_in = (taps["tok0"] if i == 0 else taps[f"x{i - 1}"])[0].contiguous()
        x_mid, x_out = taps[f"xmid{i}"][0].contiguous(), taps[f"x{i}"][0].contiguous()
        shift = tuple(m.shift_size) if i % 2 else (0, 0)
        # ---- attention branch on the oracle's layer input
        hi, lo = ops.split_pair(x_in.to(dev), kd)
        mod = F.linear(lat, att.norm.modulation.weight, att.norm.modulation.bias).contiguous()
        wq = ops.pad_cols(att.to_qkv.weight.detach(), kd, torch.bfloat16)
        o = torch.zeros(n, kd, dtype=torch.bfloat16, device=dev)
        ops.qkv_attention_fused(hi, wq, att.scale.detach().reshape(-1).float(), 1, grid, heads, shift, out=o.view(1, n, kd)[..., :d], k=d,
                                head_dim=hd)
        y = ops.gemm(o[:, :d], ops.pad_cols(att.wo.weight.detach(), kd, torch.bfloat16)[:, :d])
        ops.modnorm_residual_pair(y, hi, lo, att.norm.norm.weight.detach(), att.norm.norm.bias.detach(), mod, n, d)
        ea = rel_l2(ops.pair_value(hi, lo, d).cpu() - x_in, x_mid - x_in)
        # ---- feed-forward branch on the oracle's mid-layer stream
        hi, lo = ops.split_pair(x_mid.to(dev), kd)
        mod = F.linear(lat, ff.norm.modulation.weight, ff.norm.modulation.bias).contiguous()
        mlp = ff.w2.weight.shape[1]
        w1i = ff.w1.weight.detach().view(2, mlp, d).permute(1, 0, 2).reshape(2 * mlp, d)  # rows gate_0, up_0, gate_1, up_1, ...
        w2w = ff.w2.weight.detach()
        if mlp & 1:  # int(8/3 * 1280) = 3413: one zero (gate, up) row pair and a zero w2 column, as swift_amd/engine.py packs them
            w1i = torch.cat([w1i, w1i.new_zeros(2, d)], 0)
            w2w = torch.cat([w2w, w2w.new_zeros(d, 1)], 1)
        kh = ops.k_pad(torch.bfloat16, w2w.shape[1])
        hbuf = torch.zeros(n, kh, dtype=torch.bfloat16, device=dev)  # (k-padding columns of w2's operand must be finite: zero)
        ops.gemm(hi[:, :d], ops.pad_cols(w1i, kd, torch.bfloat16)[:, :d], out=hbuf[:, :w2w.shape[1]], epilogue=ops.EPI_SWIGLU)
        y = ops.gemm(hbuf, ops.pad_cols(w2w, kh, torch.bfloat16))
        ops.modnorm_residual_pair(y, hi, lo, ff.norm.norm.weight.detach(), ff.norm.norm.bias.detach(), mod, n, d)
        ef = rel_l2(ops.pair_value(hi, lo, d).cpu() - x_mid, x_out - x_mid)
        print(f"{name} layer {i:2d}: attention branch rel-L2 {ea:.3e}, feed-forward branch rel-L2 {ef:.3e}")
        worst["attention"], worst["feed-forward"] = max(worst["attention"], ea), max(worst["feed-forward"], ef)
    print(f"worst branch: attention {worst['attention']:.3e}, feed-forward {worst['feed-forward']:.3e}")
    # (measured: attention 7.0e-4 .. 1.2e-3, feed-forward 1.6e-4 .. 1.8e-4 -- the asked-for 1.5e-2 would not notice a 10 x regression)
    assert worst["attention"] <= 3e-3 and worst["feed-forward"] <= 5e-4


def test_bf16_engine_unit_alone_vs_in_a_batch(dev):
    """One unit evaluated alone takes the split-K wo / w2 path (two bf16 slabs summed by the norm kernel, csrc/forward.hip
    small_m_splitk); in a batch of two it takes the one-product path.  The two round the branch output differently -- at bf16
    level -- so a bf16-engine trajectory depends on how units were batched at that level and no more: the same unit, alone and
    as half of a batch, within the depth-2 bf16 bound (fp32-grade engines: bit-for-bit independent of batching)."""
    net, _ = build(SMALLB, 5, dev)
    nb = 12  # 12 units x 1024 tokens = 144 output tiles of the d-wide GEMMs: past one round of the grid, so no split-K; one unit: 12
    x, cond = det_normal((nb, 69, 64, 64), 5, "x").to(dev), det_normal((nb, 72, 64, 64), 5, "cond").to(dev)
    t = torch.linspace(0.3, 1.5, nb, device=dev)
    pick = (0, 5, 11)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        yb = net(x, t, condition=cond, auxiliary=0.6)
        y1 = torch.cat([net(x[j:j + 1], t[j:j + 1], condition=cond[j:j + 1], auxiliary=0.6) for j in pick], 0)
    with torch.no_grad():
        fb = net(x, t, condition=cond, auxiliary=0.6)
        f1 = torch.cat([net(x[j:j + 1], t[j:j + 1], condition=cond[j:j + 1], auxiliary=0.6) for j in pick], 0)
    e = rel_l2(y1.float().cpu(), yb[list(pick)].float().cpu())
    ef = rel_l2(f1.cpu(), fb[list(pick)].cpu())
    print(f"bf16 engine, unit alone (split-K wo / w2) vs inside a batch of {nb}: rel-L2 {e:.3e}; fp32 engine: {ef:.3e}")
    assert e < BF16_EMU_TOL
    assert ef < 1e-6
