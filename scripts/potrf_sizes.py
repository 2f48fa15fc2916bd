"""Factorisation time (potrf + super-block inverse merges) by size through the kernel hook."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
ctx = lp_amd.Context(0)
for m in (512, 1024, 2048, 4096):
    rng = np.random.default_rng(0)
    B = rng.standard_normal((m, m + 64))
    M = B @ B.T
    L, info, ms = ctx.k_potrf(M, repeats=10)
    err = np.abs(np.tril(L) - np.linalg.cholesky(M)).max() / np.abs(M).max() ** 0.5
    print(f"m={m}: {ms*1e3:.1f} us  info {info}  rel err vs numpy {err:.1e}", flush=True)
