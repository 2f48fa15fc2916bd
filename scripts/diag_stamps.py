import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LPIPM_DIAG_STAMPS"] = "1"
import numpy as np
import lp_amd as lp
ctx = lp.default_context(0)
for m in (128, 1024):
    rng = np.random.default_rng(0)
    B = rng.standard_normal((m, 2 * m))
    M = B @ B.T
    L, info, ms = ctx.k_potrf(M, repeats=3)
    print(m, "potrf ms", ms, "info", info, flush=True)
