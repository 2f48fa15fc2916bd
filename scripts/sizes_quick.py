"""Solve time at the three single-LP BASELINE shapes and the lockstep C4 shard, one line each (for A/B runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
o = lp_amd.InteriorPoint.default().opts()
out = []
for (m, n, reps) in ((512, 1024, 30), (1024, 2048, 20), (4096, 8192, 8)):
    A, b, c, xs = synth.planted_lp(0, m, n)
    ctx = lp_amd.Context(0); ctx.upload_arrays(A, b, c)
    for _ in range(3): ctx.solve_raw(o)
    t = time.perf_counter()
    for _ in range(reps): rc, x, fun, it, _ = ctx.solve_raw(o)
    dt = (time.perf_counter() - t) / reps
    out.append(f"{m}x{n}: {dt*1e3:.3f} ms ({it/dt:.0f} it/s)")
    ctx.close()
probs = [synth.planted_lp(s, 1024, 2048) for s in range(32)]
ctx = lp_amd.Context(0)
ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
ctx.solve_lockstep(o)
t = time.perf_counter()
for _ in range(5): res = ctx.solve_lockstep(o)
dt = (time.perf_counter() - t) / 5
out.append(f"lockstep 32x(1024x2048): {dt*1e3:.2f} ms ({32/dt:.0f} LP/s)")
print(" | ".join(out), flush=True)
