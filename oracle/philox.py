"""Oracle (test infrastructure): Philox4x32-10 + Box-Muller, NumPy restatement.

The reference draws its noise from TensorFlow's unseeded global Philox stream
(``tf.random.normal`` at ``Pyesian/optimizers/SGLD.py:67``, ``HMC.py:171``; tfp
samplers at ``BBB.py:234-237``, ``SVGD.py:154``) -- SURVEY.md Appendix A5:
bit-parity with that stream is a non-goal.  The HIP library uses its own
counter-based generator so that CPU and GPU draw the same numbers; this file
is the published Philox4x32-10 algorithm (Salmon et al., SC'11, "Parallel
random numbers: as easy as 1, 2, 3") restated for checking the device stream.

Stream definition shared with ``csrc/pyz_rng.h``:
    counter = (idx4_lo, idx4_hi, step, stream)   key = (seed_lo, seed_hi)
    element e of a tensor uses idx4 = e // 4 and output word e % 4.
    u = ((word >> 9) + 0.5) * 2**-23            (exactly representable in fp32)
    words (0,1) -> (r cos t, r sin t), words (2,3) likewise, with
    r = sqrt(-2 ln u_a), t = 2 pi u_b.
"""

from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0: int, k1: int, rounds: int = 10):
    """Vectorised Philox4x32; counters are uint32 arrays, key words Python ints."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64) + np.zeros_like(c0)
    c2 = np.asarray(c2, dtype=np.uint64) + np.zeros_like(c0)
    c3 = np.asarray(c3, dtype=np.uint64) + np.zeros_like(c0)
    for r in range(rounds):
        kk0 = np.uint64((k0 + r * _W0) & 0xFFFFFFFF)
        kk1 = np.uint64((k1 + r * _W1) & 0xFFFFFFFF)
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ kk0) & _MASK, lo1, (hi0 ^ c3 ^ kk1) & _MASK, lo0
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def words(seed: int, stream: int, step: int, n: int) -> np.ndarray:
    """The first n uint32 words of stream (seed, stream, step)."""
    n4 = (n + 3) // 4
    idx = np.arange(n4, dtype=np.uint64)
    r = philox4x32(idx & _MASK, idx >> np.uint64(32), np.uint64(step & 0xFFFFFFFF),
                   np.uint64(stream & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack(r, axis=1).reshape(-1)[:n]


def _unit(w: np.ndarray) -> np.ndarray:
    return ((w >> np.uint32(9)).astype(np.float64) + 0.5) * 2.0 ** -23


def normal(seed: int, stream: int, step: int, n: int, dtype=np.float64) -> np.ndarray:
    """n standard normals of stream (seed, stream, step), float64 Box-Muller."""
    n4 = (n + 3) // 4
    w = words(seed, stream, step, 4 * n4).reshape(n4, 4)
    out = np.empty((n4, 4), dtype=np.float64)
    for a in (0, 2):
        r = np.sqrt(-2.0 * np.log(_unit(w[:, a])))
        t = 2.0 * np.pi * _unit(w[:, a + 1])
        out[:, a] = r * np.cos(t)
        out[:, a + 1] = r * np.sin(t)
    return out.reshape(-1)[:n].astype(dtype)


def uniform(seed: int, stream: int, step: int, n: int, dtype=np.float64) -> np.ndarray:
    return _unit(words(seed, stream, step, n)).astype(dtype)
