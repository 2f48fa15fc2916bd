"""One profiled workload for rocprofv3: a few C3 (4096x8192) solves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8192)
A, b, c, xs = synth.planted_lp(0, m, n)
ctx = lp.default_context(0)
ctx.upload_arrays(A, b, c)
o = lp.InteriorPoint.default().opts()
for _ in range(3):
    rc, x, fun, its, _ = ctx.solve_raw(o)
print("rc", rc, "its", its, "err", np.abs(x - xs).max())
