#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the ACTUAL reference (stockeh/swift).

Runs only in the build container, where ``/root/reference`` is mounted.  The
reference's Python never travels to the GPU box; these small input/output
vectors do.  Usage:  python tools/make_golden.py [--only NAME]

Import recipe: the reference needs ``omegaconf``,
``hydra`` and ``ezpz`` (absent here) only for type annotations, a logger and a
``_target_`` importer, and ``h5py`` only for file reading; four import-time
stub modules stand in for them.  No reference source is copied: the modules are
imported from where they lie and driven with seeded inputs.

Weights come from ``swift_amd.utils.detinit`` (Philox + Box-Muller, not a torch
RNG stream) and are loaded into the reference modules with ``load_state_dict``,
so fixtures for big configurations hold only (seed, outputs).
"""
from __future__ import annotations

import argparse
import importlib
import logging
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/src"
OUT = os.path.join(ROOT, "tests", "golden")


def install_stubs():
    sys.path.insert(0, REF)
    om = types.ModuleType("omegaconf")
    om.ListConfig = type("ListConfig", (list,), {})
    om.DictConfig = type("DictConfig", (dict,), {})
    om.OmegaConf = type("OmegaConf", (), {})
    sys.modules["omegaconf"] = om
    ez = types.ModuleType("ezpz")
    ez.get_logger = logging.getLogger
    sys.modules["ezpz"] = ez
    hy, hu = types.ModuleType("hydra"), types.ModuleType("hydra.utils")

    def instantiate(cfg, *a, **kw):
        cfg = dict(cfg)
        tgt = cfg.pop("_target_")
        kw.pop("_convert_", None)
        kw.pop("_recursive_", None)
        mod, name = tgt.rsplit(".", 1)
        return getattr(importlib.import_module(mod), name)(*a, **{**cfg, **kw})

    hu.instantiate = instantiate
    hy.utils = hu
    sys.modules["hydra"], sys.modules["hydra.utils"] = hy, hu
    sys.modules["h5py"] = types.ModuleType("h5py")  # data/era5.py imports it; never called here


install_stubs()
from swift.data.era5 import ERA5Dataset  # noqa: E402
from swift.generating.factory import sampler_factory  # noqa: E402
from swift.models.precond import PassPrecond  # noqa: E402
from swift.models.swinv2 import SwinV2, timestep_embedding  # noqa: E402
from swift.training.loss import (  # noqa: E402
    CRPSLoss,
    SCMLoss,
    TrigFlowLoss,
    _calculate_latitude_weights,
    _calculate_variable_weights,
)

from swift_amd.utils.detinit import det_normal, state_fingerprint, swinv2_state  # noqa: E402

TINY = dict(img=(32, 64), n_vars=4, n_forc=3, window=(4, 4), shift=(2, 2), patch=(2, 2), dim=96, heads=4, depth=3)
SMALLB = dict(img=(64, 64), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1056, heads=12, depth=2)
SWIFTB = dict(img=(128, 256), n_vars=69, n_forc=3, window=(16, 16), shift=(8, 8), patch=(2, 2), dim=1056, heads=12, depth=12)


def model_cfg(c, logvar=False):
    return dict(_target_="swift.models.swinv2.SwinV2", window_size=list(c["window"]), shift_size=list(c["shift"]),
                patch_size=list(c["patch"]), depth=c["depth"], dim=c["dim"], heads=c["heads"], logvar=logvar,
                timestep_weight=1.0)


def build_ref_net(c, seed, logvar=False, sigma_data=1.0):
    nv, nf = c["n_vars"], c["n_forc"]
    net = PassPrecond(model_cfg(c, logvar), img_resolution=list(c["img"]), img_channels=nv,
                      condition_channels=nv + nf, auxiliary_dim=1, sigma_min=0, sigma_max=float("inf"),
                      sigma_data=sigma_data)
    grid = (c["img"][0] // c["patch"][0], c["img"][1] // c["patch"][1])
    state = swinv2_state(grid=grid, in_channels=2 * nv + nf, out_channels=nv, patch_size=c["patch"], depth=c["depth"],
                         dim=c["dim"], heads=c["heads"], auxiliary_dim=1, logvar=logvar, seed=seed)
    missing = net.load_state_dict(state, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.eval(), state


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")


class FakeDDP(torch.nn.Module):
    """loss.py reaches through ``net.module`` (training/loss.py:213,217)."""

    def __init__(self, m):
        super().__init__()
        self.module = m

    def forward(self, *a, **k):
        return self.module(*a, **k)


class FakeERA5(ERA5Dataset):
    """The real standardisation methods of data/era5.py on synthetic statistics (no h5 files)."""

    def __init__(self, c, seed, forc_bank):
        torch.utils.data.Dataset.__init__(self)
        nv, nf = c["n_vars"], c["n_forc"]
        self.variables = [f"v{i}" for i in range(nv)]
        self.forcings = [f"f{i}" for i in range(nf)]
        self.intervals = [6, 12, 24]
        self.residual = True
        self.x_means = det_normal((nv + nf, 1, 1), seed, "x_mean", std=2.0).numpy()
        self.x_stds = (det_normal((nv + nf, 1, 1), seed, "x_std", std=0.3).abs() + 0.5).numpy()
        self.t_stds = {d: (det_normal((nv, 1, 1), seed, f"t_std{d}", std=0.05).abs() + 0.1).numpy() for d in (6, 12, 24)}
        self.t_means = {d: np.zeros_like(self.t_stds[d]) for d in (6, 12, 24)}
        self._shape = (nv, *c["img"])
        self._bank = forc_bank  # [T, nf, H, W] physical forcings

    def get_forcings(self, idx):
        return self._bank[int(idx)].clone()


# --------------------------------------------------------------------------- fixtures


@torch.no_grad()
def fx_swinv2_tiny():
    c, seed = TINY, 1
    net, state = build_ref_net(c, seed, logvar=True)
    B, nv, nf = 2, c["n_vars"], c["n_forc"]
    x = det_normal((B, 2 * nv + nf, *c["img"]), seed, "x")
    t = torch.tensor([0.3, 1.2])
    aux = torch.tensor([[0.6], [1.2]])
    y = net.model(x, t, auxiliary=aux)
    yn = net.model(x, t, auxiliary=aux, jvp=True)
    y2, lv = net.model(x, t, auxiliary=aux, return_logvar=True)
    assert torch.equal(y, y2)
    # through PassPrecond with split (x, condition) and scalar auxiliary (precond.py:21-31,133-148)
    yp = net(x[:, :nv], t, x[:, nv:], 0.6)
    temb = timestep_embedding(torch.tensor([0.0, 1.0, math.pi / 2]), c["dim"])
    ar = torch.arange(2 * 4 * 4, dtype=torch.float32).reshape(1, 2, 4, 4)
    pe = net.model.patch_embed
    from einops import rearrange
    pat = rearrange(ar, "b c (h p1) (w p2) -> b (h w) (p1 p2 c)", p1=pe.patch_size[0], p2=pe.patch_size[1])
    save("swinv2_tiny", seed=seed, fingerprint=state_fingerprint(state), x=x, t=t, aux=aux, y_flash=y, y_naive=yn,
         logvar=lv, y_precond_scalar_aux=yp, temb=temb, patchify_arange=pat)


@torch.no_grad()
def fx_swinv2_smallb():
    c, seed = SMALLB, 2
    net, state = build_ref_net(c, seed)
    B, nv, nf = 2, c["n_vars"], c["n_forc"]
    x = det_normal((B, 2 * nv + nf, *c["img"]), seed, "x")
    t = torch.tensor([math.pi / 2, 0.7])
    aux = torch.tensor([[0.6], [2.4]])
    y = net.model(x, t, auxiliary=aux)
    yn = net.model(x, t, auxiliary=aux, jvp=True)
    # the reference's own reduced-precision path (trainer.py:191-196 runs the net under bf16 autocast): its distance
    # from its fp32 output calibrates the tolerance of the bf16 engine
    with torch.autocast("cpu", dtype=torch.bfloat16):
        yb = net.model(x, t, auxiliary=aux).float()
    save("swinv2_smallb", seed=seed, fingerprint=state_fingerprint(state), t=t, aux=aux, y_flash=y,
         bf16_autocast_rel=float((yb - y).norm() / y.norm()),
         y_naive_stats=np.array([float(yn.mean()), float(yn.std()), float((yn - y).norm() / y.norm())]))


@torch.no_grad()
def fx_attention_hd88():
    """One reference Attention module (to_qkv -> cosine attention -> wo -> ModulatedNorm) at Swift-B width."""
    seed = 3
    from swift.models.swinv2 import Attention
    att = Attention(1056, 12, 88, flash=True).eval()
    st = swinv2_state(grid=(16, 16), in_channels=4, out_channels=4, patch_size=(1, 1), depth=1, dim=1056, heads=12,
                      seed=seed, prefix="")
    att.load_state_dict({k[len("transformer.layers.0.0."):]: v for k, v in st.items()
                         if k.startswith("transformer.layers.0.0.")})
    x = det_normal((2, 256, 1056), seed, "x")
    tl = det_normal((2, 1056), seed, "t", std=0.5)
    y = att(x, tl)
    yn = att(x, tl, jvp=True)
    save("attention_hd88", seed=seed, y=y, naive_rel=float((yn - y).norm() / y.norm()))


@torch.no_grad()
def fx_samplers_tiny():
    c, seed = TINY, 4
    net, state = build_ref_net(c, seed)
    B, nv, nf = 2, c["n_vars"], c["n_forc"]
    cond = det_normal((B, nv + nf, *c["img"]), seed, "cond")
    lat = det_normal((B, nv, *c["img"]), seed, "lat")
    ren = [det_normal((B, nv, *c["img"]), seed, f"ren{i}") for i in range(4)]
    from swift.generating.diffusion import DiffusionSampler
    S = DiffusionSampler(net)

    def feeder():
        it = iter(ren)
        return lambda like: next(it).to(like)

    kw = dict(sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    out = {}
    for n in (1, 2, 3):
        out[f"scm{n}"] = S.scm_solver(lat.clone(), condition=cond, randn_like=feeder(), num_steps=n, **kw)
    out["scm_mid"] = S.scm_solver(lat.clone(), condition=cond, randn_like=feeder(), num_steps=2,
                                  intermediates=[0.9, 0.4], **kw)
    out["dpm2s3"] = S.dpm_solver_2s(lat.clone(), condition=cond, num_steps=3, **kw)
    # factory path with a torch generator (generating/factory.py:47-60): record what it drew
    g = torch.Generator().manual_seed(7)
    smp = sampler_factory("scm", net, num_steps=1, **kw)
    out["factory_scm1"] = smp(cond, generator=g)
    g = torch.Generator().manual_seed(7)
    out["factory_latents"] = torch.randn((B, nv, *c["img"]), generator=g)
    save("samplers_tiny", seed=seed, fingerprint=state_fingerprint(state), cond=cond, lat=lat,
         **{f"ren{i}": r for i, r in enumerate(ren)}, **out)


@torch.no_grad()
def fx_rollout_tiny():
    """generate.py:85-131 driven by hand with the reference's sampler and ERA5Dataset methods."""
    c, seed, steps, interval = TINY, 5, 3, 6
    net, state = build_ref_net(c, seed)
    B, nv, nf = 2, c["n_vars"], c["n_forc"]
    bank = det_normal((B + steps + 2, nf, *c["img"]), seed, "forc", std=1.5, mean=0.5)
    ds = FakeERA5(c, seed, bank)
    idx = torch.tensor([0, 2])
    X = det_normal((B, nv, *c["img"]), seed, "X0")
    sampler = sampler_factory("scm", net, num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=interval / 10.0)
    gen = torch.Generator().manual_seed(11)
    traj = [ds.unstandardize_x(X)]
    X0 = X.clone()
    for i in range(steps):
        Xc = torch.cat([X, ds.standardize_x(torch.stack([ds.get_forcings(j + int(i * interval // 6)) for j in idx], 0))], 1)
        Y = sampler(Xc, generator=gen)
        X_un = ds.unstandardize_x(Xc)[:, :nv]
        Y_un = ds.unstandardize_t(Y, delta=int(interval))
        X = X_un + Y_un
        traj.append(X)
        X = ds.standardize_x(X)
    gen = torch.Generator().manual_seed(11)
    lats = [torch.randn((B, nv, *c["img"]), generator=gen) for _ in range(steps)]
    save("rollout_tiny", seed=seed, fingerprint=state_fingerprint(state), X0=X0, idx=idx, bank=bank,
         x_mean=ds.x_means, x_std=ds.x_stds, t_std6=ds.t_stds[6], traj=torch.stack(traj, 1),
         latents=torch.stack(lats, 0))


def fx_losses_tiny():
    c, seed = TINY, 6
    net, state = build_ref_net(c, seed, logvar=True)
    net.train().requires_grad_(True)
    B, nv, nf = 2, c["n_vars"], c["n_forc"]
    bank = det_normal((B + 8, nf, *c["img"]), seed, "forc")
    ds = FakeERA5(c, seed, bank)
    # variables must be real names for the weight tables (training/loss.py:35-55)
    ds.variables = ["2m_temperature", "10m_u_component_of_wind", "geopotential_500", "temperature_850"]
    target = det_normal((B, nv, *c["img"]), seed, "target")
    cond = det_normal((B, nv + nf, *c["img"]), seed, "cond")
    aux = torch.tensor([0.6, 0.6])
    ddp = FakeDDP(net)
    gsel = ["model.head.head.0.weight", "model.transformer.layers.1.0.to_qkv.weight", "model.pos_embed",
            "model.transformer.layers.0.1.norm.modulation.weight", "model.transformer.layers.2.0.scale",
            "model.logvar_embed.weight"]
    out = {}

    def grads():
        named = dict(net.named_parameters())
        g = np.array([float(named[k].grad.norm()) if named[k].grad is not None else -1.0 for k in gsel])
        net.zero_grad(set_to_none=True)
        return g

    noise = dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0)
    # TrigFlow: draws u = rand([B,1,1,1]) then z = randn_like(x)  (loss.py:66-71,133-136)
    torch.manual_seed(21)
    L = TrigFlowLoss(ds, dict(noise), sigma_data=1.0)
    val = L(ddp, target, condition=cond, auxiliary=aux)
    val.backward()
    out["trigflow"], out["trigflow_g"] = float(val), grads()
    torch.manual_seed(21)
    out["trigflow_u"], out["trigflow_z"] = torch.rand([B, 1, 1, 1]), torch.randn_like(target)
    # sCM (loss.py:196-260)
    torch.manual_seed(22)
    L = SCMLoss(ds, dict(noise), sigma_data=1.0, tangent_warmup_kimg=3)
    val = L(ddp, target, step=1200, condition=cond, auxiliary=aux)
    val.backward()
    out["scm"], out["scm_g"] = float(val), grads()
    torch.manual_seed(22)
    out["scm_u"], out["scm_z"] = torch.rand([B, 1, 1, 1]), torch.randn_like(target)
    # CRPS multistep, steps=3, ensemble 2 (loss.py:373-445); noise order: member-major, step-minor
    steps = 3
    torch.manual_seed(23)
    L = CRPSLoss(ds, sigma_data=1.0, ensemble_size=2, alpha=0.95)
    val = L(ddp, target, condition=cond, auxiliary=aux, idx=[0, 3], steps=steps)
    val.backward()
    out["crps"], out["crps_g"] = float(val), grads()
    torch.manual_seed(23)
    out["crps_lat"] = torch.stack([torch.stack([torch.randn_like(target) for _ in range(steps)]) for _ in range(2)])
    save("losses_tiny", seed=seed, fingerprint=state_fingerprint(state), target=target, cond=cond, aux=aux, bank=bank,
         x_mean=ds.x_means, x_std=ds.x_stds, t_std6=ds.t_stds[6], grad_keys=np.array(gsel), w_lat=L.w_lat, w_var=L.w_var,
         **out)


@torch.no_grad()
def fx_swiftb_step():
    """BASELINE config 1: Swift-B, 1 member x 1 IC x 1 six-hour step, fp32 CPU."""
    c, seed = SWIFTB, 1234
    net, state = build_ref_net(c, seed)
    nv, nf = c["n_vars"], c["n_forc"]
    cond = det_normal((1, nv + nf, *c["img"]), seed, "cond")
    lat = det_normal((1, nv, *c["img"]), seed, "lat")
    from swift.generating.diffusion import DiffusionSampler
    import time
    t0 = time.time()
    y = DiffusionSampler(net).scm_solver(lat, condition=cond, num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    print(f"reference Swift-B scm step on CPU: {time.time() - t0:.2f} s ({torch.get_num_threads()} threads)")
    yb = DiffusionSampler(net).scm_solver(lat, condition=cond, num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6,
                                          denoise_dtype=torch.bfloat16).float()
    save("swiftb_step", seed=seed, fingerprint=state_fingerprint(state), y_sub=y[0, ::4, ::8, ::8],
         bf16_autocast_rel=float((yb - y).norm() / y.norm()),
         stats=np.array([float(y.mean()), float(y.std()), float(y.abs().max()), float(y.double().norm())]))


@torch.no_grad()
def fx_swiftb_long():
    """BASELINE configs[2] and [3] at full size on the reference (fp32 CPU, ~100 Swift-B evaluations, minutes):
    (a) ``dpm_solver_2s`` with solver/2s.yaml's num_steps 20 = 39 network evaluations of one sample;
    (b) a 60-step autoregressive rollout of one (member, IC) unit, generate.py:97-131 driven by hand with the reference's
        scm sampler and ERA5Dataset standardisation, latents keyed per step (det_normal), forcing bank keyed by seed.
    Outputs are stored sub-sampled ([::4, ::8, ::8]) plus the full-field norm of every step."""
    from swift.generating.diffusion import DiffusionSampler
    import time
    c, seed = SWIFTB, 1234
    net, state = build_ref_net(c, seed)
    nv, nf = c["n_vars"], c["n_forc"]
    S = DiffusionSampler(net)
    cond = det_normal((1, nv + nf, *c["img"]), seed, "cond")
    lat = det_normal((1, nv, *c["img"]), seed, "lat")
    t0 = time.time()
    y2s = S.dpm_solver_2s(lat, condition=cond, num_steps=20, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    print(f"reference dpm_solver_2s (39 evaluations): {time.time() - t0:.1f} s")
    steps, interval = 60, 6
    bank = det_normal((steps + 2, nf, *c["img"]), seed, "forc", std=1.5, mean=0.5)
    ds = FakeERA5(c, seed, bank)
    X = det_normal((1, nv, *c["img"]), seed, "X0")
    sub, norms = [ds.unstandardize_x(X)[0, ::4, ::8, ::8].clone()], [float(ds.unstandardize_x(X).double().norm())]
    t0 = time.time()
    for i in range(steps):
        Xc = torch.cat([X, ds.standardize_x(ds.get_forcings(i * interval // 6)[None])], 1)
        Y = S.scm_solver(det_normal((1, nv, *c["img"]), seed, f"lat{i}"), condition=Xc, num_steps=1, sigma_min=0.02, sigma_max=200.0,
                         auxiliary=interval / 10.0)
        Xn = ds.unstandardize_x(Xc)[:, :nv] + ds.unstandardize_t(Y, delta=interval)
        sub.append(Xn[0, ::4, ::8, ::8].clone())
        norms.append(float(Xn.double().norm()))
        X = ds.standardize_x(Xn)
        if i % 10 == 9:
            print(f"  rollout step {i + 1}/{steps}: {time.time() - t0:.0f} s")
    save("swiftb_long", seed=seed, fingerprint=state_fingerprint(state), y2s_sub=y2s[0, ::4, ::8, ::8],
         y2s_norm=float(y2s.double().norm()), traj_sub=torch.stack(sub, 0), traj_norm=np.array(norms),
         x_mean=ds.x_means, x_std=ds.x_stds, t_std6=ds.t_stds[6])



def fx_era5_tiny():
    """The reference's h5-backed ERA5Dataset / ERA5RollOutDataset (data/era5.py:11-256) over a small on-disk tree in its own
    layout (tests/era5_fixture.py writes it; the ``*.h5`` files are npz underneath and a stand-in h5py exposes the mapping
    interface the loader uses -- this image has no h5py)."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import era5_fixture as fx
    from swift.data.era5 import ERA5RollOutDataset
    sys.modules["h5py"].File = fx._File  # the reference module holds this module object
    out = {}
    with tempfile.TemporaryDirectory() as root:
        fx.write_tree(root)
        np.random.seed(0)
        ds = ERA5Dataset(root, list(fx.VARS), list(fx.FORC), intervals=[6, 12, 24], split="train", residual=True)
        out["len"], out["shape"] = len(ds), np.array(ds._shape)
        for spec in [(0, 1, 6), (3, 1, 12), (2, 2, 6), (1, 3, 12), (4, 1, 24)]:
            (x, t), (idx, delta) = ds[spec]
            tag = "_".join(map(str, spec))
            out[f"x_{tag}"], out[f"t_{tag}"], out[f"d_{tag}"] = x, t, float(delta)
        out["forc5"] = ds.get_forcings(5)
        out["time7"] = np.array(str(ds.get_time(7)))
        lat, lon = ds.get_lat_lon()
        out["lat"], out["lon"] = lat, lon
        v = det_normal((2, len(fx.VARS), *fx.SHAPE), 3, "v")
        out["unstd_t12"] = ds.unstandardize_t(v.clone(), 12)
        out["std_x_forc"] = ds.standardize_x(ds.get_forcings(2)[None])
        # (residual=False is not exercised: the reference's standardize_t indexes the statistics ARRAY with delta there,
        #  data/era5.py:161 with :95-99 -- IndexError for fewer than 7 channels, channel 6's statistics otherwise)
        ro = ERA5RollOutDataset(8, root, list(fx.VARS), list(fx.FORC), intervals=[6, 12, 24], split="train", residual=True)
        x, ts, idx = ro[1]
        out["ro_x"], out["ro_t"], out["ro_len"] = x, ts, len(ro)
    save("era5_tiny", **out)



def fx_weights_aux():
    import yaml
    with open("/root/reference/src/swift/configs/data/era5-flare-1.4.yaml") as f:
        d = yaml.safe_load(f)
    variables = d["dataset"]["variables"]
    save("weights_aux", variables=np.array(variables), w_lat128=_calculate_latitude_weights(128),
         w_var69=_calculate_variable_weights(variables), w_lat32=_calculate_latitude_weights(32))


def fx_muon_tiny():
    """Two steps of the reference's SingleDeviceMuonWithAuxAdam on seeded parameters / gradients (muon.py:267-338)."""
    from swift.training.optimizers.muon import SingleDeviceMuonWithAuxAdam
    shapes = [(96, 64), (64, 160), (128, 128), (48,), (7, 5)]
    params = [torch.nn.Parameter(det_normal(sh, 77, f"p{i}") * 0.05) for i, sh in enumerate(shapes)]
    groups = [dict(params=params[:3], use_muon=True, lr=0.02, weight_decay=0.01),
              dict(params=params[3:], use_muon=False, lr=3e-4, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-10)]
    opt = SingleDeviceMuonWithAuxAdam(groups)
    out = {f"p{i}_0": p.detach().clone().numpy() for i, p in enumerate(params)}
    for st in range(2):
        for i, p in enumerate(params):
            g = det_normal(p.shape, 78 + st, f"g{i}")
            out[f"g{i}_{st}"] = g.numpy()
            p.grad = g.clone()
        opt.step()
        for i, p in enumerate(params):
            out[f"p{i}_{st + 1}"] = p.detach().clone().numpy()
    save("muon_tiny", **out)


@torch.no_grad()
def fx_val_tiny():
    """dpm_solver (diffusion.py:289-353) alone, and the reference's RMSE_rollout (training/validate.py:23-127) driven by
    its own "dpm" and "scm" samplers on the fake dataset.  validate.py needs three more import-time stubs (mpi4py,
    torchinfo, swift.utils.io -- none of them is touched by RMSE_rollout)."""
    ti = types.ModuleType("torchinfo")
    ti.summary = lambda *a, **k: None
    sys.modules["torchinfo"] = ti
    sys.modules["swift.utils.io"] = types.ModuleType("swift.utils.io")
    mp = types.ModuleType("mpi4py")
    mp.MPI = types.SimpleNamespace(COMM_WORLD=None)
    sys.modules["mpi4py"] = mp
    sys.modules["ezpz"].get_rank = lambda: 0
    sys.modules["ezpz"].get_world_size = lambda: 1
    from swift.generating.diffusion import DiffusionSampler
    from swift.training.validate import RMSE_rollout
    c, seed, interval = TINY, 8, 8   # 8 six-hour steps = 2 days -> columns [6h, day 1, day 2]
    net, state = build_ref_net(c, seed)
    B, nv, nf = 2, c["n_vars"], c["n_forc"]
    cond = det_normal((B, nv + nf, *c["img"]), seed, "cond")
    lat = det_normal((B, nv, *c["img"]), seed, "lat")
    kw = dict(sigma_min=0.02, sigma_max=200.0, auxiliary=0.6)
    out = {"dpm4": DiffusionSampler(net).dpm_solver(lat.clone(), condition=cond, num_steps=4, **kw),
           "dpm4_noise": DiffusionSampler(net).dpm_solver(lat.clone(), condition=cond, num_steps=4, use_pp=False, **kw)}
    bank = det_normal((B + interval + 4, nf, *c["img"]), seed, "forc", std=1.5, mean=0.5)
    ds = FakeERA5(c, seed, bank)
    lat_deg = np.linspace(-88.0, 88.0, c["img"][0]).astype(np.float32)
    ds.get_lat_lon = lambda: (lat_deg, np.zeros(c["img"][1], dtype=np.float32))
    idx = [0, 3]
    X0 = det_normal((B, nv, *c["img"]), seed, "X0")
    TS = det_normal((B, interval // 4 + 1, nv, *c["img"]), seed, "TS", std=2.0)  # unstandardised targets [6h, day1, day2]
    for mode, skw in (("dpm", dict(num_steps=3)), ("scm", dict(num_steps=1))):
        smp = sampler_factory(mode, net, **skw, **kw)
        gen = torch.Generator().manual_seed(21)
        agg, sep = RMSE_rollout(smp, iter([(X0.clone(), TS.clone(), idx)]), ds, interval, torch.device("cpu"), rng=gen,
                                num_batches=1)
        out[f"agg_{mode}"], out[f"sep_{mode}"] = np.float64(agg), sep
    gen = torch.Generator().manual_seed(21)
    out["latents"] = torch.stack([torch.randn((B, nv, *c["img"]), generator=gen) for _ in range(interval)], 0)
    save("val_tiny", seed=seed, fingerprint=state_fingerprint(state), cond=cond, lat=lat, X0=X0, TS=TS, idx=np.array(idx),
         bank=bank, lat_deg=lat_deg, x_mean=ds.x_means, x_std=ds.x_stds, t_std6=ds.t_stds[6], **out)


@torch.no_grad()
def fx_metrics_tiny():
    """The reference's offline ensemble metrics (eval/metrics.py:39-134) on a seeded 5-member ensemble (xarray stubbed:
    only its main() touches it)."""
    sys.modules.setdefault("xarray", types.ModuleType("xarray"))
    from swift.eval.metrics import lat_weighted_crps, lat_weighted_rmse, lat_weighted_spread_skill_ratio
    B, N, V, H, W = 3, 5, 4, 16, 32
    y = det_normal((B, V, H, W), 91, "y")
    pred = y[:, None] * 0.8 + det_normal((B, N, V, H, W), 91, "p") * 0.5
    lat = np.linspace(-87.1875, 87.1875, H).astype(np.float64)
    names = [f"v{i}" for i in range(V)]
    out = {}
    for fn in (lat_weighted_rmse, lat_weighted_crps, lat_weighted_spread_skill_ratio):
        out.update({k: float(v) for k, v in fn(pred, y, names, lat, "6h").items()})
    save("metrics_tiny", pred=pred, y=y, lat=lat, keys=np.array(sorted(out)), values=np.array([out[k] for k in sorted(out)]))



def fx_trainer_tiny():
    """The reference's optimisation step itself -- ``Trainer._backward_step`` (training/trainer.py:199-247): LR warm-up /
    cosine schedule, ``nan_to_num`` gradient sanitising, AdamW step, EMA rule -- called unbound on a stand-in ``self`` (the
    constructor needs ezpz / DDP; the step does not).  Gradients are injected through a loss that is linear in the
    parameters, NaN / +-inf entries included.  Import-time stubs: xarray, mpi4py, torchinfo, swift.utils.io, ezpz.History."""
    import torch._dynamo  # noqa: F401  (the optimiser imports it lazily, and its module scan trips over spec-less stubs)
    for name in ("xarray", "torchinfo"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchinfo"].summary = lambda *a, **k: None
    sys.modules["xarray"].Dataset = type("Dataset", (), {})  # a return annotation of Trainer.train
    sys.modules.setdefault("swift.utils.io", types.ModuleType("swift.utils.io"))
    mp = types.ModuleType("mpi4py")
    mp.MPI = types.SimpleNamespace(COMM_WORLD=None)
    sys.modules.setdefault("mpi4py", mp)
    ez = sys.modules["ezpz"]
    ez.get_rank, ez.get_world_size = (lambda: 0), (lambda: 1)
    ez.History = type("History", (), {})
    from swift.training.trainer import Trainer
    torch.manual_seed(0)
    shapes = {"model.pos_embed": (1, 6, 8), "model.layers.0.w1.weight": (16, 8), "model.layers.0.norm.norm.weight": (8,),
              "model.layers.0.norm.norm.bias": (8,), "model.layers.0.norm.modulation.weight": (16, 8), "model.head.weight": (4, 8)}
    net = torch.nn.ParameterList([torch.nn.Parameter(det_normal(sh, 12, k, std=0.3)) for k, sh in shapes.items()])  # keeps order
    ema = torch.nn.ParameterList([torch.nn.Parameter(v.detach().clone() + 0.01, requires_grad=False) for v in net])
    names = list(shapes)
    no_decay = [i for i, n in enumerate(names) if "pos_embed" in n or ("norm" in n and "modulation" not in n)]  # train.py:275-286
    params = list(net.parameters())
    groups = [{"params": [p for i, p in enumerate(params) if i not in no_decay], "weight_decay": 0.05},
              {"params": [params[i] for i in no_decay], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=2e-3, betas=(0.9, 0.95), eps=1e-6)
    me = types.SimpleNamespace(lr_rampup_kimg=0.004, optimizer=opt, base_lr=[g["lr"] for g in opt.param_groups], lr_min_factor=0.01,
                               lr_cosine_anneal=True, total_kimg=0.02, scaler=torch.amp.GradScaler("cpu", enabled=False), net=net,
                               ema=ema, ema_halflife_kimg=0.5, ema_rampup_ratio=0.05, global_batch_size=2)
    out = {"names": np.array(names), "no_decay": np.array(no_decay)}
    for k, v in zip(names, params):
        out[f"p0_{k}"] = v.detach().clone()
    for k, v in zip(names, ema.parameters()):
        out[f"e0_{k}"] = v.detach().clone()
    nimgs = [2, 4, 10, 20]          # warm-up (2 of 4), boundary, mid-cosine, end
    for st, nimg in enumerate(nimgs):
        opt.zero_grad(set_to_none=True)
        G = [det_normal(p.shape, 100 + st, n, std=0.5) for n, p in zip(names, params)]
        G[1].view(-1)[3] = float("nan")
        G[1].view(-1)[5] = float("inf")
        G[5].view(-1)[0] = float("-inf")
        G[2].view(-1)[1] = float("nan")
        loss = sum((p * g).sum() for p, g in zip(params, G))
        Trainer._backward_step(me, nimg, loss)
        out[f"lr_{st}"] = np.array([g["lr"] for g in opt.param_groups])
        for n, p, e, g in zip(names, params, ema.parameters(), G):
            out[f"g{st}_{n}"], out[f"p{st + 1}_{n}"], out[f"e{st + 1}_{n}"] = g, p.detach().clone(), e.detach().clone()
    save("trainer_tiny", nimgs=np.array(nimgs), **out)


def fx_index_streams():
    """Index streams of the reference's data samplers (data/samplers.py:9-85): ``InfiniteSampler`` over several (dataset size,
    rank, world, seed, window, offset) settings -- more than two laps each, so the carried-over swaps and the rank phase
    across laps are covered -- and ``DeltaBatchSampler`` batches on top of it."""
    from swift.data.samplers import DeltaBatchSampler, InfiniteSampler
    import itertools
    cases = [  # n, rank, world, shuffle, seed, window, offset, count
        (37, 0, 1, True, 0, 0.5, 1, 120), (37, 1, 3, True, 5, 0.5, 1, 60), (37, 2, 3, True, 5, 0.5, 4, 60),
        (64, 3, 8, True, 11, 0.25, 2, 40), (10, 0, 2, False, 0, 0.5, 3, 30), (5, 0, 1, True, 1, 0.2, 1, 25),
        (101, 7, 8, True, 2024, 1.0, 5, 50)]
    out = {"cases": np.array(cases, dtype=np.float64)}
    for c, (n, rank, world, shuffle, seed, window, offset, count) in enumerate(cases):
        s = InfiniteSampler(range(n), rank=rank, num_replicas=world, shuffle=shuffle, seed=seed, window_size=window)
        if offset > 1:
            s.set_offset(offset)
        items = list(itertools.islice(iter(s), count))
        out[f"stream_{c}"] = np.array([i[0] if isinstance(i, tuple) else i for i in items], dtype=np.int64)
        assert all((isinstance(i, tuple) and i[1] == offset) == (offset > 1) for i in items)
    s = InfiniteSampler(range(50), rank=1, num_replicas=2, shuffle=True, seed=3)
    s.set_offset(3)
    b = DeltaBatchSampler(s, 4, [6, 12, 24], seed=3)
    out["delta_batches"] = np.array(list(itertools.islice(iter(b), 12)), dtype=np.int64)  # [12, 4, (index, offset, delta)]
    save("index_streams", **out)


def fx_scm_distill_tiny():
    """SCMLoss with ``distillation=True`` and a v-prediction teacher (training/loss.py:204-208): dx_t/dt comes from
    ``sigma_d * net_pretrained(x_t / sigma_d, t, condition, auxiliary)`` instead of ``cos t z - sin t x``.  Student weights:
    seed 6 (as losses_tiny), teacher weights: seed 16, same tiny architecture (no logvar head on the teacher)."""
    c, seed = TINY, 6
    net, state = build_ref_net(c, seed, logvar=True)
    teacher, tstate = build_ref_net(c, seed + 10, logvar=False)
    net.train().requires_grad_(True)
    B, nv, nf = 2, c["n_vars"], c["n_forc"]
    ds = FakeERA5(c, seed, det_normal((B + 8, nf, *c["img"]), seed, "forc"))
    ds.variables = ["2m_temperature", "10m_u_component_of_wind", "geopotential_500", "temperature_850"]
    target = det_normal((B, nv, *c["img"]), seed, "target")
    cond = det_normal((B, nv + nf, *c["img"]), seed, "cond")
    aux = torch.tensor([0.6, 0.6])
    gsel = ["model.head.head.0.weight", "model.transformer.layers.1.0.to_qkv.weight", "model.pos_embed",
            "model.transformer.layers.0.1.norm.modulation.weight", "model.transformer.layers.2.0.scale",
            "model.logvar_embed.weight"]
    torch.manual_seed(24)
    L = SCMLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), sigma_data=1.0, tangent_warmup_kimg=3,
                distillation=True)
    val = L(FakeDDP(net), target, step=1200, condition=cond, auxiliary=aux, net_pretrained=teacher)
    val.backward()
    named = dict(net.named_parameters())
    g = np.array([float(named[k].grad.norm()) for k in gsel])
    assert all(p.grad is None for p in teacher.parameters())
    torch.manual_seed(24)
    u, z = torch.rand([B, 1, 1, 1]), torch.randn_like(target)
    # the same draw without a teacher, for scale: distillation must change the loss
    net.zero_grad(set_to_none=True)
    torch.manual_seed(24)
    plain = SCMLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), sigma_data=1.0, tangent_warmup_kimg=3)(
        FakeDDP(net), target, step=1200, condition=cond, auxiliary=aux)
    save("scm_distill_tiny", seed=seed, teacher_seed=seed + 10, fingerprint=state_fingerprint(state),
         teacher_fingerprint=state_fingerprint(tstate), target=target, cond=cond, aux=aux, u=u, z=z, loss=float(val),
         loss_without_teacher=float(plain), grad_keys=np.array(gsel), grad_norms=g, w_lat=L.w_lat, w_var=L.w_var)


def fx_swiftb_2s_bf16():
    """How far the reference's OWN bf16 path is from its fp32 path on BASELINE configs[2] at full size: ``dpm_solver_2s``
    (39 Swift-B evaluations) with ``denoise_dtype=torch.bfloat16`` (CPU autocast) against the fp32 output stored in
    swiftb_long.npz, on the stored sub-sample -- the yardstick for the bf16 engine's bound in tests/test_gpu_model.py."""
    from swift.generating.diffusion import DiffusionSampler
    import time
    c, seed = SWIFTB, 1234
    net, state = build_ref_net(c, seed)
    nv, nf = c["n_vars"], c["n_forc"]
    S = DiffusionSampler(net)
    cond = det_normal((1, nv + nf, *c["img"]), seed, "cond")
    lat = det_normal((1, nv, *c["img"]), seed, "lat")
    t0 = time.time()
    y16 = S.dpm_solver_2s(lat, condition=cond, num_steps=20, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6,
                          denoise_dtype=torch.bfloat16)
    print(f"reference dpm_solver_2s under bf16 autocast: {time.time() - t0:.1f} s")
    ref = torch.from_numpy(np.load(os.path.join(OUT, "swiftb_long.npz"))["y2s_sub"])
    sub = y16[0, ::4, ::8, ::8].float()
    rel = float((sub.double() - ref.double()).norm() / ref.double().norm())
    print(f"bf16-autocast vs fp32 (sub-sample): rel-L2 {rel:.3e}")
    save("swiftb_2s_bf16", seed=seed, fingerprint=state_fingerprint(state), bf16_autocast_rel_sub=rel)


def fx_swiftb_fp64():
    """The fp64 truth for BASELINE configs[1] and [2] at full size: the reference itself with ``net.double()``, fp64 inputs
    and fp64 default dtype (so the solvers' time grids and ``timestep_embedding`` are fp64 too) -- the same seeded weights
    and inputs as swiftb_step / swiftb_long, which are fp32 values and upcast exactly.  Stored: the sub-sampled fp64
    outputs and how far the reference's OWN fp32 run (the stored swiftb_step / swiftb_long sub-samples) is from them.
    The yardstick that tells fp32 rounding noise from a defect in tests/test_gpu_model.py."""
    from swift.generating.diffusion import DiffusionSampler
    import time
    c, seed = SWIFTB, 1234
    net, state = build_ref_net(c, seed)
    nv, nf = c["n_vars"], c["n_forc"]
    cond = det_normal((1, nv + nf, *c["img"]), seed, "cond").double()
    lat = det_normal((1, nv, *c["img"]), seed, "lat").double()
    torch.set_default_dtype(torch.float64)
    try:
        net = net.double()
        S = DiffusionSampler(net)
        t0 = time.time()
        with torch.no_grad():
            y1 = S.scm_solver(lat, condition=cond, num_steps=1, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6,
                              denoise_dtype=torch.float64)
            print(f"reference fp64 scm step: {time.time() - t0:.1f} s", flush=True)
            assert y1.dtype == torch.float64
            t0 = time.time()
            y2 = S.dpm_solver_2s(lat, condition=cond, num_steps=20, sigma_min=0.02, sigma_max=200.0, auxiliary=0.6,
                                 denoise_dtype=torch.float64)
            print(f"reference fp64 dpm_solver_2s (39 evaluations): {time.time() - t0:.1f} s", flush=True)
            assert y2.dtype == torch.float64
    finally:
        torch.set_default_dtype(torch.float32)
    y1s, y2s = y1[0, ::4, ::8, ::8], y2[0, ::4, ::8, ::8]
    r1 = torch.from_numpy(np.load(os.path.join(OUT, "swiftb_step.npz"))["y_sub"]).double()
    r2 = torch.from_numpy(np.load(os.path.join(OUT, "swiftb_long.npz"))["y2s_sub"]).double()
    d1 = float((r1 - y1s).norm() / y1s.norm())
    d2 = float((r2 - y2s).norm() / y2s.norm())
    print(f"reference fp32 vs fp64 (sub-sample): scm step {d1:.3e}, dpm_solver_2s {d2:.3e}")
    save("swiftb_fp64", seed=seed, fingerprint=state_fingerprint(state), y1_sub=y1s, y2s_sub=y2s,
         ref_fp32_vs_fp64_step=d1, ref_fp32_vs_fp64_2s=d2,
         y1_norm=float(y1.norm()), y2s_norm=float(y2.norm()))


ALL = dict(swiftb_fp64=fx_swiftb_fp64, swiftb_2s_bf16=fx_swiftb_2s_bf16, scm_distill_tiny=fx_scm_distill_tiny, index_streams=fx_index_streams, era5_tiny=fx_era5_tiny, swiftb_long=fx_swiftb_long, trainer_tiny=fx_trainer_tiny, metrics_tiny=fx_metrics_tiny, val_tiny=fx_val_tiny, muon_tiny=fx_muon_tiny, swinv2_tiny=fx_swinv2_tiny, swinv2_smallb=fx_swinv2_smallb, attention_hd88=fx_attention_hd88,
           samplers_tiny=fx_samplers_tiny, rollout_tiny=fx_rollout_tiny, losses_tiny=fx_losses_tiny,
           swiftb_step=fx_swiftb_step, weights_aux=fx_weights_aux)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    for k, fn in ALL.items():
        if a.only is None or a.only == k:
            print("==", k)
            fn()
