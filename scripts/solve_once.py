"""A few solves of one planted LP (for rocprofv3 traces): solve_once.py m n reps  (environment knobs apply)."""
import os, sys
os.environ["LPIPM_EXPERIMENTAL"] = "1"      # the library reads its measurement knobs only with the master switch on
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth
m, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A, b, c, xs = synth.planted_lp(0, m, n)
cx = lp.Context(0)
cx.upload_arrays(A, b, c)
o = lp.InteriorPoint.default().opts()
for _ in range(reps):
    rc, x, fun, it, _ = cx.solve_raw(o)
print("rc", rc, "it", it, "err", np.abs(x - xs).max())
