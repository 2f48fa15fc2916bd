"""Pass times over A at the BASELINE shapes: gemv_n, gemv_t (1 vector) and the one-read dual pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
for m, n in ((512, 1024), (1024, 2048), (4096, 8192)):
    A, b, c, _ = synth.planted_lp(0, m, n)
    ctx = lp_amd.Context(0); ctx.upload_arrays(A, b, c)
    w, v = np.random.default_rng(0).standard_normal(n), np.random.default_rng(1).standard_normal(m)
    tn = ctx.k_gemv_n(w, repeats=20)[1]; tt = ctx.k_gemv_t(v, repeats=20)[1]; td = ctx.k_gemv_dual(w, v, repeats=20)[2]
    gb = 8.0 * m * n / 1e9
    print(f"{m}x{n}: gemv_n {tn*1e3:.1f} us ({gb/tn:.0f} GB/s)  gemv_t(+reduce) {tt*1e3:.1f} us  dual {td*1e3:.1f} us ({gb/td:.0f} GB/s)", flush=True)
    ctx.close()
