"""Deterministic synthetic weights / fields that do not depend on any torch RNG stream.

There is no network for checkpoints or ERA5 data, so tests, ``bench.py`` and
``smoke()`` run on random-init weights of the reference architecture.  The
draws come from numpy's Philox counter generator + an explicit Box-Muller, so
the very same numbers can be re-created in the build container (where the
golden fixtures are produced from the reference) and on the GPU box.

State-dict keys and shapes follow the reference's ``PassPrecond(SwinV2)``
(models/swinv2.py:278-303, models/precond.py:123-131).
Unlike the reference's init, *nothing is left at zero / one*: the reference
zero-initialises ``modulation`` and ``head`` (output would be identically 0)
and its biases, and sets LayerNorm to (1, 0); parity tests need every term to
carry signal, so those get small random values too.
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch


def det_normal(shape, seed: int, tag: str, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    """N(mean, std^2) float32 tensor, a pure function of (shape, seed, tag)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.array([seed & 0xFFFFFFFFFFFFFFFF, zlib.crc32(tag.encode())], dtype=np.uint64)
    rng = np.random.Generator(np.random.Philox(key=key))
    m = (n + 1) // 2
    u1 = rng.random(m)
    u2 = rng.random(m)
    r = np.sqrt(-2.0 * np.log1p(-u1))  # u1 in [0,1) -> 1-u1 in (0,1]
    z = np.concatenate([r * np.cos(2 * math.pi * u2), r * np.sin(2 * math.pi * u2)])[:n]
    return torch.from_numpy((z * std + mean).astype(np.float32).reshape(shape))


def swinv2_state(
    *,
    grid,
    in_channels: int,
    out_channels: int,
    patch_size,
    depth: int,
    dim: int,
    heads: int,
    auxiliary_dim: int = 1,
    logvar: bool = False,
    seed: int = 0,
    prefix: str = "model.",
) -> dict:
    """Deterministic PassPrecond(SwinV2) state dict (166 tensors for Swift-B)."""
    gh, gw = grid
    p1, p2 = patch_size
    d = dim
    hd = dim // heads
    mlp = int(8 / 3.0 * dim)
    pf, po = in_channels * p1 * p2, out_channels * p1 * p2

    def w(name, *shape, std=0.02, mean=0.0):
        return det_normal(shape, seed, name, std=std, mean=mean)

    s = {}
    s["pos_embed"] = w("pos_embed", 1, gh * gw, d)
    s["patch_embed.emb.weight"] = w("patch_embed.emb.weight", d, pf)
    s["patch_embed.emb.bias"] = w("patch_embed.emb.bias", d)
    for l in ("l1", "l2"):
        s[f"latent_embed.{l}.weight"] = w(f"latent_embed.{l}.weight", d, d)
        s[f"latent_embed.{l}.bias"] = w(f"latent_embed.{l}.bias", d)
    if logvar:
        s["logvar_embed.weight"] = w("logvar_embed.weight", 1, d)
        s["logvar_embed.bias"] = w("logvar_embed.bias", 1)
    if auxiliary_dim:
        s["auxiliary_embed.weight"] = w("auxiliary_embed.weight", d, auxiliary_dim)
        s["auxiliary_embed.bias"] = w("auxiliary_embed.bias", d)
    for i in range(depth):
        a, f = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1."
        sc = w(a + "scale", 1, heads, 1, 1, std=0.3, mean=math.log(10.0))
        sc[0, (i + 1) % heads, 0, 0] = 5.0  # one head beyond the ln(100) clamp (swinv2.py:125)
        s[a + "scale"] = sc
        for pre in (a, f):
            s[pre + "norm.norm.weight"] = w(pre + "norm.norm.weight", d, std=0.1, mean=1.0)
            s[pre + "norm.norm.bias"] = w(pre + "norm.norm.bias", d, std=0.1)
            s[pre + "norm.modulation.weight"] = w(pre + "norm.modulation.weight", 2 * d, d)
            s[pre + "norm.modulation.bias"] = w(pre + "norm.modulation.bias", 2 * d)
        s[a + "to_qkv.weight"] = w(a + "to_qkv.weight", 3 * hd * heads, d)
        s[a + "wo.weight"] = w(a + "wo.weight", d, hd * heads)
        s[f + "w1.weight"] = w(f + "w1.weight", 2 * mlp, d)
        s[f + "w2.weight"] = w(f + "w2.weight", d, mlp)
    s["head.head.0.weight"] = w("head.head.0.weight", po, d)
    return {prefix + k: v for k, v in s.items()}


def state_fingerprint(state: dict) -> float:
    """Order-independent float64 checksum used by fixtures to detect generator drift."""
    tot = 0.0
    for k in sorted(state):
        v = state[k].double().flatten()
        idx = torch.arange(1, v.numel() + 1, dtype=torch.float64)
        tot += float((v * torch.cos(idx)).sum()) * (1 + (zlib.crc32(k.encode()) % 97) / 97.0)
    return tot
