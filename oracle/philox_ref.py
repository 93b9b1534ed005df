"""Oracle: Philox4x32-10 counter RNG + Box-Muller normals (test infrastructure only).

The reference draws its noise with ``torch.randn(shape, device)`` (models/utils/helpers.py:37-40,
models/diffusion/ddpm.py:216,244); that stream is device/library specific, so the product's
in-kernel noise uses the published Philox4x32-10 algorithm (Salmon et al., "Parallel Random
Numbers: As Easy as 1, 2, 3", SC'11; Random123 v1.14 ``philox4x32_R(10, ...)``) instead.  This
numpy restatement is pinned by the Random123 known-answer vectors in tests/test_philox.py.

Layout contract with csrc/diffusion.hip: one Philox call per 4 consecutive elements,
counter = (idx & 0xffffffff, idx >> 32, step, stream), key = (seed_lo, seed_hi), idx = element // 4;
u1 = ((r0 >> 8) + 0.5) * 2^-24, u2 = ((r1 >> 8) + 0.5) * 2^-24,
z0 = sqrt(-2 ln u1) cos(2 pi u2), z1 = sqrt(-2 ln u1) sin(2 pi u2); (r2, r3) -> (z2, z3) likewise.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32, key: (..., 2) uint32 -> (..., 4) uint32."""
    c0, c1, c2, c3 = (np.asarray(ctr[..., i], dtype=np.uint32).copy() for i in range(4))
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        for r in range(10):
            if r > 0:
                k0 = (k0 + W0).astype(np.uint32)
                k1 = (k1 + W1).astype(np.uint32)
            p0 = c0.astype(np.uint64) * M0
            p1 = c2.astype(np.uint64) * M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _LO).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _LO).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
    return np.stack([c0, c1, c2, c3], axis=-1)


def _u01(r):
    return ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def philox_normal(n, seed, step, stream=0):
    """The n fp32 normals the device kernel produces for (seed, step, stream)."""
    nblk = (n + 3) // 4
    idx = np.arange(nblk, dtype=np.uint64)
    ctr = np.stack([(idx & _LO).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                    np.full(nblk, step, dtype=np.uint32), np.full(nblk, stream, dtype=np.uint32)], axis=-1)
    key = np.stack([np.full(nblk, seed & 0xFFFFFFFF, dtype=np.uint32),
                    np.full(nblk, (seed >> 32) & 0xFFFFFFFF, dtype=np.uint32)], axis=-1)
    r = philox4x32_10(ctr, key)
    out = np.empty((nblk, 4), dtype=np.float32)
    two_pi = np.float32(2.0 * np.pi)
    for j in (0, 2):
        u1, u2 = _u01(r[:, j]), _u01(r[:, j + 1])
        rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        out[:, j] = rad * np.cos(two_pi * u2)
        out[:, j + 1] = rad * np.sin(two_pi * u2)
    return out.reshape(-1)[:n]
