"""Parity sweep: many seeded planted LPs, HIP path vs the C oracle: iteration-count agreement and |dx|_inf."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth
from oracle import capi as oracle
ctx = lp.default_context(0)
o = lp.InteriorPoint.default().opts()
rows = []
for (m, n) in [(32, 64), (64, 200), (128, 256), (200, 333), (256, 512), (384, 1000), (512, 1024), (640, 1100), (1024, 2048)]:
    for seed in range(6 if m <= 512 else 3):
        A, b, c, xs = synth.planted_lp(seed, m, n)
        ctx.upload_arrays(A, b, c)
        rc, x, fun, it, _ = ctx.solve_raw(o)
        ref = oracle.solve(A, b, c, want_log=False)
        dx = np.abs(x - ref["x_slack"]).max() if rc == 0 and ref["status"] == 0 else float("nan")
        rows.append((m, n, seed, rc, ref["status"], it, ref["iterations"], dx, np.abs(ref["x_slack"] - xs).max()))
        print("%4d x %4d seed %d  status %d/%d  iterations %2d/%2d  |x_gpu-x_oracle| %.2e  |x_oracle-x*| %.2e" % rows[-1], flush=True)
r = np.array(rows, dtype=float)
print("cases", len(r), "status agree", int((r[:, 3] == r[:, 4]).sum()), "iterations agree", int((r[:, 5] == r[:, 6]).sum()),
      "max |dx|", np.nanmax(r[:, 7]), "median |dx|", np.nanmedian(r[:, 7]))
