"""Leak / stability soak: contexts created and destroyed, uploads of changing geometry, single / lockstep / batch solves;
free device memory must come back to where it started."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lp_amd
from lp_amd import synth
def free_mb():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(0); return f / 2**20
o = lp_amd.InteriorPoint.default().opts()
base = None
for rnd in range(6):
    ctx = lp_amd.Context(0)
    for (m, n) in ((64, 160), (300, 700), (64, 160), (1024, 2048), (130, 400)):
        A, b, c, xs = synth.planted_lp(rnd, m, n)
        ctx.upload_arrays(A, b, c)
        for _ in range(3):
            rc, x, fun, it, _ = ctx.solve_raw(o); assert rc == 0 and np.abs(x - xs).max() < 1e-5
    probs = [synth.planted_lp(s, 96, 200) for s in range(6)]
    ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
    res = ctx.solve_lockstep(o); assert all(r[0] == 0 for r in res)
    mixed = [p[:3] + (0.0,) for p in probs] + [synth.planted_lp(9, 50, 120)[:3] + (0.0,)]
    out = ctx.solve_batch(mixed, o); assert all(r[0] == 0 for r in out)
    ctx.close()
    f = free_mb()
    if base is None: base = f
    print(f"round {rnd}: free device memory {f:.0f} MiB (delta vs round 0: {f - base:+.0f} MiB)", flush=True)
assert abs(free_mb() - base) < 64, "device memory leak"
print("soak ok")
