"""Per-call factorisation times at one size (spots one-off costs such as the first use of a kernel).  usage: potrf_repeat.py m [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
m = int(sys.argv[1]); calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = lp_amd.Context(0)
rng = np.random.default_rng(0)
B = rng.standard_normal((m, m + 64)); M = B @ B.T
ts = []
for _ in range(calls):
    L, info, ms = ctx.k_potrf(M, repeats=1)
    ts.append(ms * 1e3)
print(f"m={m}: " + " ".join(f"{t:.0f}" for t in ts))
