import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
m = int(sys.argv[1])
ctx = lp_amd.Context(0)
rng = np.random.default_rng(0)
B = rng.standard_normal((m, m + 64)); M = B @ B.T
for _ in range(3):
    L, info, ms = ctx.k_potrf(M, repeats=1)
    time.sleep(0.05)
print(ms)
