"""Device log vs a committed golden log, column by column.  usage: python scripts/golden_log_diff.py planted_4096x8192_s2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
name = sys.argv[1]
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".npz"))
dims, seed = name.split("_")[1], int(name.split("_s")[1]); m, n = (int(v) for v in dims.split("x"))
A, b, c, xs = synth.planted_lp(seed, m, n)
ctx = lp_amd.Context(0); ctx.upload_arrays(A, b, c)
rc, x, fun, it, rows = ctx.solve_raw(lp_amd.InteriorPoint.default().opts(), want_log=True)
print(name, "rc", rc, "it", it, int(g["iterations"]), "|x-x_gold|", np.abs(x - g["x_slack"]).max(), "|x-x*|", np.abs(x - xs).max(), "gold |x-x*|", float(g["xstar_err"]))
np.set_printoptions(linewidth=200, precision=4)
got, exp = np.array(rows), g["log"]
for k in range(len(got)):
    print(k + 1, "gpu ", got[k][:6]); print("  gold", exp[k][:6])
