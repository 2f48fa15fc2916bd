"""BASELINE config C5 shape (m=16384, n=32768: A 4 GiB, M 2 GiB) on ONE MI355X: 288 GB of HBM hold it whole.
No oracle at this size: checks are the planted vertex, primal feasibility and the indicators."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 32768)
t = time.time(); A, b, c, xs = synth.planted_lp(0, m, n); print(f"generated {m}x{n} in {time.time()-t:.1f} s", flush=True)
ctx = lp.default_context(0)
t = time.time(); ctx.upload_arrays(A, b, c); print(f"upload {time.time()-t:.2f} s", flush=True)
ctx.set_profiling(True)
o = lp.InteriorPoint.default().opts()
t = time.time(); rc, x, fun, its, rows = ctx.solve_raw(o, want_log=True); dt = time.time() - t
pt = ctx.phase_times()
print(f"solve rc={rc} iterations={its} {dt:.3f} s  {its/dt:.2f} it/s  max|x-x*|={np.abs(x-xs).max():.2e}", flush=True)
print({k: round(v / max(its, 1), 3) for k, v in pt.items() if k.endswith('_ms')}, "ms per iteration", flush=True)
fl = m * (m + 1.0) * n
print(f"A.D.A^T: {pt['adat_ms']/pt['adat_launches']:.2f} ms/launch = {fl/(pt['adat_ms']/pt['adat_launches']*1e-3)/1e12:.1f} TFLOP/s", flush=True)
r = A @ x - b
print(f"|Ax-b|_inf={np.abs(r).max():.2e}  min x={x.min():.2e}  last indicators rho_p={rows[-1][1]:.1e} rho_d={rows[-1][2]:.1e} rho_A={rows[-1][3]:.1e}", flush=True)
