"""For chosen C4 members: distance of the device's and the oracle's solution to the planted vertex, the oracle's noise floor
and the dtau margin.  usage: python scripts/c4_over.py 6 32 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c4_members.npz"))
ctx = lp_amd.Context(0)
o = lp_amd.InteriorPoint.default().opts()
for s in [int(a) for a in sys.argv[1:]]:
    A, b, c, xs = synth.planted_lp(s, int(g["m"]), int(g["n"]))
    ctx.upload_arrays(A, b, c)
    rc, x, fun, it, _ = ctx.solve_raw(o)
    xo = g["x_slack"][s]
    print(f"member {s}: it dev/oracle {it}/{int(g['iterations'][s])}  |x_dev - x_or| {np.abs(x - xo).max():.2e}  |x_dev - x*| {np.abs(x - xs).max():.2e}  "
          f"|x_or - x*| {float(g['xstar_err'][s]):.2e}  floor {float(g['floor'][s]):.2e}  margin {float(g['dtau_margin'][s]):.1e}  "
          f"c.x dev-or {c @ x - c @ xo:.2e}  |Ax-b| {np.abs(A @ x - b).max():.1e}", flush=True)
