"""GPU vs oracle on awkward shapes: tiny, ragged, very wide, tall (rank-deficient by shape), single row / column."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
from oracle import capi as oracle
ctx = lp_amd.Context(0)
o = lp_amd.InteriorPoint.default().opts()
bad = 0
shapes = [(1, 1), (1, 2), (1, 17), (2, 3), (3, 1000), (8, 100000), (17, 33), (127, 129), (128, 129), (129, 130), (129, 257),
          (255, 511), (257, 300), (300, 301), (511, 512), (640, 641), (1025, 1100)]
for (m, n) in shapes:
    if m < n:
        A, b, c, xs = synth.planted_lp(m * 31 + n, m, n)
    else:
        rng = np.random.default_rng(m); A = rng.standard_normal((m, n)); x0 = rng.uniform(0.5, 1.5, n); b = A @ x0; c = rng.uniform(0.1, 1, n)
    ref = oracle.solve(A, b, c)
    ctx.upload_arrays(A, b, c)
    rc, x, fun, it, _ = ctx.solve_raw(o)
    dx = np.abs(x - ref["x_slack"]).max() if (rc == 0 and ref["status"] == 0) else float("nan")
    ok = rc == ref["status"] and (rc != 0 or (it == ref["iterations"] and dx <= 1e-6))
    bad += not ok
    print(f"{m:5d} x {n:6d}: gpu status {rc} it {it} | oracle status {ref['status']} it {ref['iterations']} | |dx| {dx:.2e} {'OK' if ok else 'MISMATCH'}", flush=True)
# tall / square systems (m >= n): A.D.A^T is singular by shape -> both must report a numerical problem
for (m, n) in ((5, 3), (130, 129), (200, 200)):
    rng = np.random.default_rng(m * 7 + n); A = rng.standard_normal((m, n)); x0 = rng.uniform(0.5, 1.5, n); b = A @ x0; c = rng.uniform(0.1, 1, n)
    ref = oracle.solve(A, b, c)
    ctx.upload_arrays(A, b, c)
    rc, x, fun, it, _ = ctx.solve_raw(o)
    print(f"{m:5d} x {n:6d} (m >= n): gpu status {rc} it {it} | oracle status {ref['status']} it {ref['iterations']}", flush=True)
print("mismatches:", bad)
