"""Lockstep vs one-at-a-time on the same LPs: per-LP |dx|, error vs the planted optimum and the quality of both
answers as LP solutions (primal residual, objective gap) -- tells rounding sensitivity of an LP from a defect."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
m, n, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
probs = [synth.planted_lp(s, m, n) for s in range(s0, s0 + K)]
o = lp_amd.InteriorPoint.default().opts()
ctx = lp_amd.Context(0)
single = []
for A, b, c, xs in probs:
    ctx.upload_arrays(A, b, c)
    single.append(ctx.solve_raw(o, want_log=True))
ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
res = ctx.solve_lockstep(o)
worst = 0
for i, (A, b, c, xs) in enumerate(probs):
    st, x, fun, it = res[i]
    x1 = single[i][1]
    d = np.abs(x - x1).max()
    worst = max(worst, d)
    if d > 1e-6 or i < 2:
        rp = lambda v: np.abs(A @ v - b).max()
        print(f"seed {s0+i}: it single {single[i][3]} lock {it}; |x_lock-x_single| {d:.2e}; err vs x*: single {np.abs(x1-xs).max():.2e} lock {np.abs(x-xs).max():.2e}; "
              f"|Ax-b|: single {rp(x1):.2e} lock {rp(x):.2e} x* {rp(xs):.2e}; c.x - c.x*: single {c@x1-c@xs:.2e} lock {c@x-c@xs:.2e}; "
              f"last log row single {single[i][4][-1][1:4]}", flush=True)
print("worst |x_lock - x_single|", worst)
