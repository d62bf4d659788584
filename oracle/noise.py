"""CPU restatement of the counter-based latent-noise stream (TEST INFRASTRUCTURE).

The reference draws latents with ``torch.randn(shape, generator=g)`` (generating/factory.py:52-56) from one torch generator
per member consumed in batch order (generate.py:83); SURVEY.md section 7 asks the build for a counter-based stream keyed by
(member, IC, step) instead, so that a unit's noise does not depend on the sharding.  That stream is this module's subject:
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- the Random123 library,
not vendored anywhere in /root/reference; constants and round function restated from the paper) followed by Box-Muller.

Pinned by the published known-answer vectors of Random123's ``kat_vectors`` for philox4x32-10 (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter: np.ndarray, key) -> np.ndarray:
    """counter uint32 [..., 4], key (k0, k1) -> uint32 [..., 4]; ten rounds, key bumped by the Weyl constants per round."""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        h0, l0, h1, l1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [h1 ^ c[1] ^ np.uint64(k0), l1, h0 ^ c[3] ^ np.uint64(k1), l0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, -1).astype(np.uint32)


def unit_bits(seed: int, step: int, n: int) -> np.ndarray:
    """The generator's raw words for one unit: element e = word (e % 4) of counter (e // 4, 0, step lo, step hi)."""
    n4 = (n + 3) // 4
    ctr = np.zeros((n4, 4), dtype=np.uint32)
    idx = np.arange(n4, dtype=np.uint64)
    ctr[:, 0], ctr[:, 1] = (idx & MASK).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2], ctr[:, 3] = np.uint32(step & 0xFFFFFFFF), np.uint32((step >> 32) & 0xFFFFFFFF)
    seed &= 0xFFFFFFFFFFFFFFFF
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)).reshape(-1)[:n]


def unit_noise(seed: int, step: int, n: int) -> np.ndarray:
    """N(0, 1) float32 [n] for (seed, step): words (w0, w1) and (w2, w3) of a counter are two Box-Muller pairs over 24-bit
    uniforms u = ((w >> 8) + 0.5) 2^-24: z = sqrt(-2 ln u_a) * (cos 2 pi u_b, sin 2 pi u_b).  Evaluated in float64 and rounded:
    the device's fp32 logf / sincospif agree to a few ulp."""
    w = unit_bits(seed, step, (n + 3) // 4 * 4).reshape(-1, 4)
    u = ((w >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    ra, rb = np.sqrt(-2.0 * np.log(u[:, 0])), np.sqrt(-2.0 * np.log(u[:, 2]))
    z = np.stack([ra * np.cos(2 * np.pi * u[:, 1]), ra * np.sin(2 * np.pi * u[:, 1]),
                  rb * np.cos(2 * np.pi * u[:, 3]), rb * np.sin(2 * np.pi * u[:, 3])], -1)
    return z.reshape(-1)[:n].astype(np.float32)
