"""CPU restatement of the reference SwinV2 denoiser (TEST INFRASTRUCTURE).

Functional: weights come in as a ``dict[str, Tensor]`` that uses the reference
state-dict key names *below* the ``model.`` prefix of ``PassPrecond`` (i.e. the
keys of ``SwinV2.state_dict()``: ``pos_embed``, ``patch_embed.emb.weight`` ...,
see SURVEY.md section 8b).  Everything is written with explicit index maps
instead of einops/roll so that it doubles as the specification of the address
arithmetic the HIP kernels implement.

Reference: models/swinv2.py (line numbers cited per function).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class SwinCfg:
    """Constructor surface of the reference ``SwinV2`` (models/swinv2.py:255-270)."""

    img_resolution: tuple
    in_channels: int
    out_channels: int
    window_size: tuple
    shift_size: tuple
    patch_size: tuple
    depth: int = 6
    dim: int = 512
    heads: int = 12
    auxiliary_dim: int = 0
    flash: bool = True
    logvar: bool = False
    timestep_weight: float = 1.0

    @property
    def grid(self):
        return (self.img_resolution[0] // self.patch_size[0], self.img_resolution[1] // self.patch_size[1])

    @property
    def head_dim(self):
        return self.dim // self.heads

    @property
    def mlp_dim(self):
        # models/swinv2.py:160  int(8 / 3.0 * dim)
        return int(8 / 3.0 * self.dim)


SWIFT_B = dict(window_size=(16, 16), shift_size=(8, 8), patch_size=(2, 2), depth=12, dim=1056, heads=12)


# --------------------------------------------------------------------------- init


def init_state(cfg: SwinCfg, seed: int = 0, randomize_zero_init: bool = True) -> dict:
    """Seeded weights with the reference's shapes/keys (models/swinv2.py:278-303).

    The reference zero-initialises every ``modulation`` and ``head`` Linear, which
    makes the network output identically zero; for parity work those are
    re-drawn N(0, 0.02^2) when ``randomize_zero_init``.
    """
    g = torch.Generator().manual_seed(seed)

    def tn(*shape, std=0.02):
        w = torch.empty(*shape)
        torch.nn.init.trunc_normal_(w, std=std, generator=g)
        return w

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    d, gh, gw = cfg.dim, *cfg.grid
    pf = cfg.in_channels * cfg.patch_size[0] * cfg.patch_size[1]
    po = cfg.out_channels * cfg.patch_size[0] * cfg.patch_size[1]
    s = {
        "pos_embed": rn(1, gh * gw, d),
        "patch_embed.emb.weight": tn(d, pf),
        "patch_embed.emb.bias": torch.zeros(d),
        "latent_embed.l1.weight": tn(d, d),
        "latent_embed.l1.bias": torch.zeros(d),
        "latent_embed.l2.weight": tn(d, d),
        "latent_embed.l2.bias": torch.zeros(d),
    }
    if cfg.logvar:
        s["logvar_embed.weight"] = tn(1, d)
        s["logvar_embed.bias"] = torch.zeros(1)
    if cfg.auxiliary_dim:
        s["auxiliary_embed.weight"] = tn(d, cfg.auxiliary_dim)
        s["auxiliary_embed.bias"] = torch.zeros(d)
    zero_or_rand = (lambda *sh: rn(*sh)) if randomize_zero_init else (lambda *sh: torch.zeros(*sh))
    for i in range(cfg.depth):
        a, f = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1."
        s[a + "scale"] = torch.log(10 * torch.ones(1, cfg.heads, 1, 1))
        s[a + "norm.norm.weight"] = torch.ones(d)
        s[a + "norm.norm.bias"] = torch.zeros(d)
        s[a + "norm.modulation.weight"] = zero_or_rand(2 * d, d)
        s[a + "norm.modulation.bias"] = torch.zeros(2 * d)
        s[a + "to_qkv.weight"] = tn(3 * cfg.head_dim * cfg.heads, d)
        s[a + "wo.weight"] = tn(d, cfg.head_dim * cfg.heads)
        s[f + "norm.norm.weight"] = torch.ones(d)
        s[f + "norm.norm.bias"] = torch.zeros(d)
        s[f + "norm.modulation.weight"] = zero_or_rand(2 * d, d)
        s[f + "norm.modulation.bias"] = torch.zeros(2 * d)
        s[f + "w1.weight"] = tn(2 * cfg.mlp_dim, d)
        s[f + "w2.weight"] = tn(d, cfg.mlp_dim)
    s["head.head.0.weight"] = zero_or_rand(po, d)
    return s


# --------------------------------------------------------------------------- pieces


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10_000) -> torch.Tensor:
    """[sin(t f) | cos(t f)] with f_i = exp(-ln(max_period) i / half).

    models/swinv2.py:44-60 builds [cos|sin] and then swaps the halves, so the
    net layout is sin first.  Odd ``dim`` appends a zero column *before* the
    swap in the reference; that case (never used by Swift configs) is restated
    literally.
    """
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=t.dtype) / half).to(t.device)
    ang = t[:, None].to(t.dtype) * freqs[None]
    if dim % 2 == 0:
        return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    emb = torch.cat([torch.cos(ang), torch.sin(ang), torch.zeros_like(ang[:, :1])], dim=-1)
    return emb.reshape(emb.shape[0], 2, -1).flip(1).reshape(emb.shape)


def patchify(x: torch.Tensor, patch: tuple) -> torch.Tensor:
    """[B,C,H,W] -> [B, gh*gw, p1*p2*C]; feature index = (i1*p2 + i2)*C + c.

    models/swinv2.py:223-229 ("b c (h p1) (w p2) -> b (h w) (p1 p2 c)").
    """
    B, C, H, W = x.shape
    p1, p2 = patch
    gh, gw = H // p1, W // p2
    x = x.reshape(B, C, gh, p1, gw, p2).permute(0, 2, 4, 3, 5, 1)  # b gh gw p1 p2 c
    return x.reshape(B, gh * gw, p1 * p2 * C)


def unpatchify(y: torch.Tensor, patch: tuple, grid: tuple) -> torch.Tensor:
    """[B, gh*gw, C*p1*p2] -> [B,C,H,W]; feature index = (c*p1 + i1)*p2 + i2.

    models/swinv2.py:241-243 ("b (h w) (c p1 p2) -> b c (h p1) (w p2)").
    """
    B = y.shape[0]
    p1, p2 = patch
    gh, gw = grid
    C = y.shape[-1] // (p1 * p2)
    y = y.reshape(B, gh, gw, C, p1, p2).permute(0, 3, 1, 4, 2, 5)  # b c gh p1 gw p2
    return y.reshape(B, C, gh * p1, gw * p2)


def window_token_index(grid: tuple, window: tuple, shift: tuple) -> torch.Tensor:
    """Token gather map of roll(-shift) followed by window_partition.

    Returns int64 ``[nW, wh*ww]``: entry (w, j) is the index, in the un-rolled
    row-major token grid, of the j-th token of window w.  roll(x, -s)[p] =
    x[(p + s) mod n] (models/swinv2.py:193-198, 17-28).  The inverse
    (window_reverse + roll(+s), :203-208) is a scatter through the same map.
    """
    gh, gw = grid
    wh, ww = window
    sh, sw = shift
    wy = torch.arange(gh // wh).view(-1, 1, 1, 1)
    wx = torch.arange(gw // ww).view(1, -1, 1, 1)
    iy = torch.arange(wh).view(1, 1, -1, 1)
    ix = torch.arange(ww).view(1, 1, 1, -1)
    gy = (wy * wh + iy + sh) % gh
    gx = (wx * ww + ix + sw) % gw
    return (gy * gw + gx).reshape(-1, wh * ww)


def modulated_norm(x, t_lat, p, prefix, eps=1e-6):
    """LayerNorm(eps=1e-6, affine) then x*(1+scale)+shift (models/swinv2.py:77-86).

    ``x`` [B, n, d]; ``t_lat`` [B, d] (one modulation per sample; the reference
    repeats it per window, :184, which is the same numbers).
    """
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), p[prefix + "norm.weight"], p[prefix + "norm.bias"], eps)
    mod = F.linear(t_lat, p[prefix + "modulation.weight"], p[prefix + "modulation.bias"])
    scale, shift = mod[:, :d], mod[:, d:]
    return x * (1 + scale[:, None, :]) + shift[:, None, :]


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round to the nearest bf16 value (ties to even), keep the fp32 container: the operand rounding of a bf16 matrix
    product whose accumulation stays fp32 (torch autocast on the reference side, bf16 MFMA on the product side)."""
    return x.to(torch.bfloat16).to(torch.float32)


def linear_bf16(x, w, b=None):
    """F.linear with both operands rounded to bf16 and fp32 accumulation (what ``autocast(bf16)`` makes of models/swinv2.py's
    nn.Linear calls, minus the rounding of the OUTPUT, which callers apply where the product stores bf16)."""
    return F.linear(bf16_round(x), bf16_round(w), b)


def cosine_window_attention(qkv, scale, heads, naive: bool = False, emulate_bf16: bool = False):
    """qkv [Bw, n, heads*3*hd] with per-head channel blocks [q|k|v] -> [Bw, n, heads*hd].

    models/swinv2.py:119-136: q = normalize(q)*exp(min(scale, ln 100)),
    k = normalize(k), softmax(q k^T) v with softmax scale 1.0; no mask, no bias.

    ``emulate_bf16``: the norm and the softmax stay fp32 (as under autocast), the operands of the two contractions are
    rounded to bf16 -- q-hat*tau, k-hat, v, and the un-normalised probabilities e = exp(S - offset), whose row sum is taken
    over the ROUNDED values (the matrix pipe sums what it multiplies).  The offset is the row maximum (True) or, with
    ``emulate_bf16="offset0"``, zero for heads whose logit bound tau = exp(min(scale, ln 100)) is <= 48 (|S| <= tau because
    q-hat, k-hat are unit vectors, so exp cannot overflow) and the row maximum otherwise: two equally valid roundings of the
    same formula -- softmax is shift-invariant -- whose distance from each other is the noise floor of ANY bf16
    implementation of this network (tests/test_gpu_model.py calibrates the bf16 engine's bound on it).
    """
    Bw, n, c3 = qkv.shape
    hd = c3 // (3 * heads)
    qkv = qkv.reshape(Bw, n, heads, 3 * hd).permute(0, 2, 1, 3)  # b h n 3hd
    q, k, v = qkv[..., :hd], qkv[..., hd : 2 * hd], qkv[..., 2 * hd :]
    tau = torch.clamp(scale, max=math.log(1.0 / 0.01)).exp()
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12) * tau
    k = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    if emulate_bf16:
        s = bf16_round(q) @ bf16_round(k).transpose(-2, -1)
        off = s.amax(dim=-1, keepdim=True)
        if emulate_bf16 == "offset0":
            off = torch.where(tau <= 48.0, torch.zeros_like(tau), off)
        e = bf16_round((s - off).exp())
        o = (e @ bf16_round(v)) / e.sum(dim=-1, keepdim=True)
    elif naive:
        o = (q @ k.transpose(-2, -1)).softmax(dim=-1) @ v
    else:
        o = F.scaled_dot_product_attention(q, k, v, scale=1.0)
    return o.permute(0, 2, 1, 3).reshape(Bw, n, heads * hd)


def latent_embedding(cfg: SwinCfg, p: dict, t: torch.Tensor, auxiliary: Optional[torch.Tensor]):
    """t [B] (+ aux [B, aux_dim]) -> latent [B, d] (models/swinv2.py:316-321, 67-74)."""
    emb = timestep_embedding(t * cfg.timestep_weight, cfg.dim)
    if cfg.auxiliary_dim and auxiliary is not None:
        emb = emb + F.linear(
            auxiliary * math.sqrt(cfg.auxiliary_dim), p["auxiliary_embed.weight"], p["auxiliary_embed.bias"]
        )
    h = F.silu(F.linear(emb, p["latent_embed.l1.weight"], p["latent_embed.l1.bias"]))
    return F.silu(F.linear(h, p["latent_embed.l2.weight"], p["latent_embed.l2.bias"]))


def transformer_layer(cfg: SwinCfg, p: dict, i: int, x, lat, jvp: bool = False, taps: Optional[dict] = None,
                      emulate_bf16: bool = False):
    """One (attention, feed-forward) pair on the residual stream x [B, n, d].

    models/swinv2.py:186-212 (layer loop), :105-139 (attention), :89-102 (SwiGLU).

    ``emulate_bf16``: the four large Linears take bf16-rounded operands and (like an autocast nn.Linear) return a bf16-rounded
    branch output ``y``; LayerNorm, the modulation, the residual stream and its additions stay fp32.
    """
    if emulate_bf16:
        return _transformer_layer_bf16(cfg, p, i, x, lat, taps, emulate_bf16)
    B, n, d = x.shape
    a, f = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1."
    shift = cfg.shift_size if (any(cfg.shift_size) and i % 2 != 0) else (0, 0)
    idx = window_token_index(cfg.grid, cfg.window_size, shift).to(x.device)  # [nW, wn]
    nW, wn = idx.shape

    qkv = F.linear(x, p[a + "to_qkv.weight"])  # per-token, order-independent
    qkv_w = qkv[:, idx.reshape(-1)].reshape(B * nW, wn, -1)
    o_w = cosine_window_attention(qkv_w, p[a + "scale"], cfg.heads, naive=(jvp or not cfg.flash))
    o = torch.empty(B, n, o_w.shape[-1], dtype=o_w.dtype, device=x.device)
    o[:, idx.reshape(-1)] = o_w.reshape(B, nW * wn, -1)
    if taps is not None:
        taps[f"qkv{i}"] = qkv
        taps[f"attn{i}"] = o
    y = modulated_norm(F.linear(o, p[a + "wo.weight"]), lat, p, a + "norm.")
    x = x + y
    if taps is not None:
        taps[f"xmid{i}"] = x

    h = F.linear(x, p[f + "w1.weight"])
    m = cfg.mlp_dim
    h = F.silu(h[..., :m]) * h[..., m:]
    y = modulated_norm(F.linear(h, p[f + "w2.weight"]), lat, p, f + "norm.")
    return x + y


def _transformer_layer_bf16(cfg: SwinCfg, p: dict, i: int, x, lat, taps, mode=True):
    """transformer_layer with the roundings a bf16 matrix pipe applies (see there); same index maps."""
    B, n, d = x.shape
    a, f = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1."
    shift = cfg.shift_size if (any(cfg.shift_size) and i % 2 != 0) else (0, 0)
    idx = window_token_index(cfg.grid, cfg.window_size, shift).to(x.device)
    nW, wn = idx.shape
    qkv = linear_bf16(x, p[a + "to_qkv.weight"])  # fp32 accumulators: the cosine norm reads them unrounded
    qkv_w = qkv[:, idx.reshape(-1)].reshape(B * nW, wn, -1)
    o_w = cosine_window_attention(qkv_w, p[a + "scale"], cfg.heads, emulate_bf16=mode)
    o = torch.empty(B, n, o_w.shape[-1], dtype=o_w.dtype, device=x.device)
    o[:, idx.reshape(-1)] = o_w.reshape(B, nW * wn, -1)
    if taps is not None:
        taps[f"qkv{i}"] = qkv
        taps[f"attn{i}"] = o
    y = bf16_round(linear_bf16(o, p[a + "wo.weight"]))
    x = x + modulated_norm(y, lat, p, a + "norm.")
    if taps is not None:
        taps[f"xmid{i}"] = x  # the residual stream between the two branches (per-layer teacher forcing in the GPU tests)
    h = linear_bf16(x, p[f + "w1.weight"])
    m = cfg.mlp_dim
    h = F.silu(h[..., :m]) * h[..., m:]
    y = bf16_round(linear_bf16(h, p[f + "w2.weight"]))
    return x + modulated_norm(y, lat, p, f + "norm.")


def swinv2_forward(
    cfg: SwinCfg,
    p: dict,
    x: torch.Tensor,
    t: torch.Tensor,
    auxiliary: Optional[torch.Tensor] = None,
    jvp: bool = False,
    return_logvar: bool = False,
    taps: Optional[dict] = None,
    emulate_bf16: bool = False,
):
    """models/swinv2.py:305-330.  x [B,Cin,H,W], t [B] or scalar, aux [B,aux_dim].

    ``emulate_bf16`` (TEST YARDSTICK for the bf16 engine, not a reference code path): the operands of the patch embedding, of
    the four Linears per layer, of QK^T / PV and of the head are rounded to bf16, every accumulation, LayerNorm, softmax,
    the time-embedding MLP, the modulation Linears and the residual stream stay fp32.  It is what the reference's
    ``autocast(bfloat16)`` computes, with the fp32 islands a hand-written bf16 engine keeps (small Linears, patch-embedding
    and head outputs)."""
    lin = linear_bf16 if emulate_bf16 else F.linear
    tok = lin(patchify(x, cfg.patch_size), p["patch_embed.emb.weight"], p["patch_embed.emb.bias"])
    tok = tok + p["pos_embed"]
    if t.dim() == 0 or (t.dim() == 1 and t.size(0) == 1):
        t = t.repeat(tok.size(0))
    lat = latent_embedding(cfg, p, t, auxiliary)
    if taps is not None:
        taps["tok0"] = tok
        taps["lat"] = lat
    for i in range(cfg.depth):
        tok = transformer_layer(cfg, p, i, tok, lat, jvp=jvp, taps=taps, emulate_bf16=emulate_bf16)
        if taps is not None:
            taps[f"x{i}"] = tok
    out = unpatchify(lin(tok, p["head.head.0.weight"]), cfg.patch_size, cfg.grid)
    if cfg.logvar and return_logvar:
        lv = F.linear(lat, p["logvar_embed.weight"], p["logvar_embed.bias"]).squeeze(-1)
        return out, lv
    return out


# --------------------------------------------------------------------------- precond


def process_auxiliary(auxiliary, auxiliary_dim: int, batch: int, device):
    """models/precond.py:21-31."""
    if auxiliary_dim == 0:
        return None
    if auxiliary is None:
        return torch.zeros([1, auxiliary_dim], device=device)
    if not isinstance(auxiliary, torch.Tensor):
        auxiliary = torch.tensor(auxiliary, device=device)
    if auxiliary.dim() == 0 or (auxiliary.dim() == 1 and auxiliary.size(0) == 1):
        auxiliary = auxiliary.repeat(batch)
    return auxiliary.reshape(-1, auxiliary_dim)


class OracleNet:
    """``PassPrecond``-shaped callable over the functional oracle (models/precond.py:101-151).

    ``state`` uses PassPrecond key names (``model.`` prefix) so the same dict
    loads into the reference, the oracle and the product module.
    """

    def __init__(self, cfg: SwinCfg, state: dict, img_channels: int, condition_channels: int,
                 sigma_data: float = 1.0, sigma_min: float = 0.0, sigma_max: float = float("inf")):
        self.cfg = cfg
        self.p = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
        self.img_channels = img_channels
        self.condition_channels = condition_channels
        self.img_resolution = tuple(cfg.img_resolution)
        self.auxiliary_dim = cfg.auxiliary_dim
        self.sigma_data, self.sigma_min, self.sigma_max = sigma_data, sigma_min, sigma_max
        self.emulate_bf16 = False  # tests set it to get the bf16-operand yardstick through samplers and rollouts

    def __call__(self, x, t, condition=None, auxiliary=None, **kw):
        if self.emulate_bf16:
            kw.setdefault("emulate_bf16", self.emulate_bf16)
        aux = process_auxiliary(auxiliary, self.auxiliary_dim, x.size(0), x.device)
        arg = x
        if condition is not None and self.condition_channels > 0:
            arg = torch.cat([arg, condition], dim=1)
        return swinv2_forward(self.cfg, self.p, arg, t.flatten(), auxiliary=aux, **kw)
