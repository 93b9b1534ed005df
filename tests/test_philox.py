"""Philox4x32-10 oracle against the Random123 known-answer vectors (kat_vectors, philox4x32 10 rounds)."""
import numpy as np

from oracle.philox_ref import philox4x32_10, philox_normal

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_kat():
    for ctr, key, want in KAT:
        got = philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))[0]
        assert tuple(int(v) for v in got) == want


def test_philox_normal_moments():
    z = philox_normal(1 << 20, seed=1234, step=7, stream=3).astype(np.float64)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2
    # distinct (step, stream) -> distinct streams
    assert not np.allclose(z[:64], philox_normal(64, 1234, 8, 3))
    assert not np.allclose(z[:64], philox_normal(64, 1234, 7, 4))
