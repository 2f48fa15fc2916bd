"""Synthetic dense LPs for tests and bench: the C generator of liblpipm.so
(lp_amd/csrc/synth.cpp) plus a pure-numpy mirror of the same stream for cross-checking small sizes."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _capi

_M64 = (1 << 64) - 1


def planted_lp(seed: int, m: int, n: int):
    """-> (A[m,n], b[m], c[n], xstar[n]) from lpipm_synth_planted_lp (host C++)."""
    A, b, c, xs = np.empty((m, n)), np.empty(m), np.empty(n), np.empty(n)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = _capi.lib().lpipm_synth_planted_lp(seed, m, n, p(A), p(b), p(c), p(xs))
    if rc != _capi.OK:
        raise ValueError(f"lpipm_synth_planted_lp failed: {_capi.strerror(rc)}")
    return A, b, c, xs


class _Rng:
    """splitmix64 -> xoshiro256**, Box-Muller; scalar Python, for small cross-checks only."""

    def __init__(self, seed):
        z, s = seed & _M64, []
        for _ in range(4):
            z = (z + 0x9E3779B97F4A7C15) & _M64
            r = z
            r = ((r ^ (r >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            r = ((r ^ (r >> 27)) * 0x94D049BB133111EB) & _M64
            s.append(r ^ (r >> 31))
        self.s, self.spare = s, None

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & _M64

    def next(self):
        s = self.s
        result = (self._rotl((s[1] * 5) & _M64, 7) * 9) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def u01(self):
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)

    def normal(self):
        if self.spare is not None:
            v, self.spare = self.spare, None
            return v
        u1, u2 = 1.0 - self.u01(), self.u01()
        r, th = math.sqrt(-2.0 * math.log(u1)), 6.283185307179586476925286766559 * u2
        self.spare = r * math.sin(th)
        return r * math.cos(th)


def planted_lp_py(seed: int, m: int, n: int):
    """Pure-Python mirror of synth.cpp (same draw order)."""
    g = _Rng(seed)
    A = np.array([g.normal() for _ in range(m * n)]).reshape(m, n)
    g.spare = None
    perm = list(range(n))
    for i in range(m):
        j = i + g.next() % (n - i)
        perm[i], perm[j] = perm[j], perm[i]
    xs, zs = np.zeros(n), np.zeros(n)
    inB = np.zeros(n, dtype=bool)
    for i in range(m):
        xs[perm[i]] = 1.0 + g.u01()
        inB[perm[i]] = True
    ys = np.array([g.normal() for _ in range(m)])
    for j in range(n):
        if not inB[j]:
            zs[j] = 1.0 + g.u01()
    return A, A @ xs, A.T @ ys + zs, xs
