"""Repeated solves over shapes that go through every A.D.A^T decomposition (whole-tile units, several chunks per tile with the
last-arriver combine, non-uniform tail chunks, the round-2 kernel's few-tile case) and through the look-ahead of the
factorisation: every repeat must be bit-identical to the first, the timeout word of the wait kernels must stay clear, and the
result must be the planted optimum.  usage: soak_units.py [seconds=120]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
o = lp_amd.InteriorPoint.default().opts()
shapes = [(512, 1024), (1000, 5000), (1024, 2048), (2048, 4096), (1500, 9000), (4096, 8192), (3000, 3500), (640, 20000)]
ctx = lp_amd.Context(0)
t0 = time.time()
total = 0
rnd = 0
while time.time() - t0 < budget:
    for (m, n) in shapes:
        A, b, c, xs = synth.planted_lp(rnd, m, n)
        ctx.upload_arrays(A, b, c)
        first = None
        for rep in range(4):
            rc, x, fun, it, _ = ctx.solve_raw(o)
            assert rc == 0, (m, n, rnd, rc)
            if first is None:
                first = (x.copy(), it)
                assert np.abs(x - xs).max() < 2e-3, (m, n, rnd, np.abs(x - xs).max())     # (distance to the planted vertex at tol 1e-8: instance-dependent)
            else:
                assert it == first[1] and np.array_equal(x, first[0]), (m, n, rnd, rep)
            total += 1
        if time.time() - t0 > budget: break
    rnd += 1
    print(f"round {rnd}: {total} solves, {time.time() - t0:.0f} s", flush=True)
ctx.close()
print("soak ok:", total, "solves")
