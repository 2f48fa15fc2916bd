"""Resident-input lockstep batch (BASELINE config C4 shard: 32 LPs of 1024x2048 on one GPU) -- the thing to profile."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m_ = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n_ = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
probs = [synth.planted_lp(s, m_, n_) for s in range(K)]
o = lp.InteriorPoint.default().opts()
ctx = lp.Context(0)
t = time.perf_counter()
ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
tu = time.perf_counter() - t
res = ctx.solve_lockstep(o)
t = time.perf_counter()
for _ in range(reps):
    res = ctx.solve_lockstep(o)
dt = (time.perf_counter() - t) / reps
its = sum(r[3] for r in res)
err = max(np.abs(r[1] - p[3]).max() for r, p in zip(res, probs))
print(f"lockstep {K} x ({m_}x{n_}): upload {tu*1e3:.1f} ms; solve {dt*1e3:.2f} ms = {K/dt:.1f} LP/s, {its/dt:.0f} it/s "
      f"({its/K:.2f} it/LP, {dt*1e3/(its/K):.3f} ms per lockstep iteration); max err vs planted optimum {err:.2e}", flush=True)
