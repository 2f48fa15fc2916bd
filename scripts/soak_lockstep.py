"""Lockstep shard soak: the C4 shard (two half-batch views on two host threads) solved again and again; every repeat must
return the bits of the first, every member the planted optimum.  usage: soak_lockstep.py [seconds=120]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
o = lp_amd.InteriorPoint.default().opts()
t0 = time.time()
total = 0
for (K, m, n) in ((32, 1024, 2048), (18, 200, 520), (40, 512, 1024)):
    probs = [synth.planted_lp(s, m, n) for s in range(K)]
    ctx = lp_amd.Context(0)
    ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
    first = ctx.solve_lockstep(o)
    assert all(r[0] == 0 for r in first)
    assert max(np.abs(r[1] - p[3]).max() for r, p in zip(first, probs)) < 1e-3
    t1 = time.time()
    while time.time() - t1 < budget / 3:
        res = ctx.solve_lockstep(o)
        for a, b in zip(first, res):
            assert a[3] == b[3] and np.array_equal(a[1], b[1])
        total += K
    ctx.close()
    print(f"{K} x ({m}x{n}): ok, {total} member solves so far, {time.time() - t0:.0f} s", flush=True)
print("lockstep soak ok:", total, "member solves")
