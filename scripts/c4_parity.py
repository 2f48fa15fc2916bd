"""All committed C4 members (tests/golden/c4_members.npz: 1024x2048, seeds 0..K-1) through the single-LP path and
the lockstep path (chunks of 32), each against the oracle's x.  Prints per-path statistics and the members beyond
max(1e-6, 10*floor).   usage: python scripts/c4_parity.py [K] [single|lock|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c4_members.npz"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else len(g["seeds"])
which = sys.argv[2] if len(sys.argv) > 2 else "both"
m, n = int(g["m"]), int(g["n"])
o = lp_amd.InteriorPoint.default().opts()
ctx = lp_amd.Context(0)
err = {"single": np.full(K, np.nan), "lock": np.full(K, np.nan)}
its = {"single": np.zeros(K, int), "lock": np.zeros(K, int)}
xerr = {"single": np.full(K, np.nan), "lock": np.full(K, np.nan)}
for s0 in range(0, K, 32):
    seeds = list(range(s0, min(K, s0 + 32)))
    probs = [synth.planted_lp(s, m, n) for s in seeds]
    if which in ("single", "both"):
        for s, (A, b, c, xs) in zip(seeds, probs):
            ctx.upload_arrays(A, b, c)
            rc, x, fun, it, _ = ctx.solve_raw(o)
            assert rc == 0
            err["single"][s] = np.abs(x - g["x_slack"][s]).max(); its["single"][s] = it; xerr["single"][s] = np.abs(x - xs).max()
    if which in ("lock", "both"):
        ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
        for s, (st, x, fun, it), pr in zip(seeds, ctx.solve_lockstep(o), probs):
            assert st == 0
            err["lock"][s] = np.abs(x - g["x_slack"][s]).max(); its["lock"][s] = it; xerr["lock"][s] = np.abs(x - pr[3]).max()
    print("done", s0, flush=True)
bar = np.maximum(1e-6, 10 * g["floor"][:K])
for k in ("single", "lock"):
    e = err[k]
    if np.isnan(e).all(): continue
    bad = np.where(e > bar)[0]
    mm = np.where(its[k] != g['iterations'][:K])[0]
    print(f"{k}: iteration mismatches {[(int(s_), int(its[k][s_]), int(g['iterations'][s_]), float(g['floor'][s_])) for s_ in mm]}; |x-x_oracle| median {np.median(e):.2e} max {e.max():.2e}; "
          f"> 1e-6: {(e > 1e-6).sum()}; > max(1e-6,10*floor): {len(bad)} {bad.tolist()}")
    print(f"   err vs planted x*: median {np.median(xerr[k]):.2e}; oracle's own {np.median(g['xstar_err'][:K]):.2e}; "
          f"members where {k} is further from x* than the oracle: {(xerr[k] > g['xstar_err'][:K]).sum()} of {K}")
    for s in np.where(e > 1e-6)[0]:
        print(f"   seed {s}: err {e[s]:.2e} floor {g['floor'][s]:.2e} x*err {k} {xerr[k][s]:.2e} oracle {g['xstar_err'][s]:.2e}")
