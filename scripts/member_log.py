"""Per-iteration log of one planted LP on the device path next to the oracle's (indicators.rs:8-23 columns).
usage: python scripts/member_log.py <seed> [m n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
from oracle import capi as oracle
seed = int(sys.argv[1]); m, n = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 2048)
A, b, c, xs = synth.planted_lp(seed, m, n)
ctx = lp_amd.Context(0); ctx.upload_arrays(A, b, c)
rc, x, fun, it, rows = ctx.solve_raw(lp_amd.InteriorPoint.default().opts(), want_log=True)
r = oracle.solve(A, b, c)
print(f"seed {seed}: gpu {it} iterations (rc {rc}), oracle {r['iterations']}; |x_gpu - x*| {np.abs(x - xs).max():.2e}, |x_oracle - x*| {np.abs(r['x_slack'] - xs).max():.2e}")
for k in range(max(it, r["iterations"])):
    g = rows[k] if k < len(rows) else None
    o = r["log"][k] if k < len(r["log"]) else None
    f = lambda t: " ".join(f"{v:10.3e}" for v in t[:6]) if t else " " * 65
    print(f"{k+1:2d} gpu {f(g)} | oracle {f(o)}")
