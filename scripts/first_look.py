import time, numpy as np, sys
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lp_amd as lp
from lp_amd import synth
ctx = lp.default_context(0)
for it in (2000, 20000):
    print("mfma probe", it, ctx.k_mfma_f64_probe(it), flush=True)
for (m, n) in [(512, 1024), (1024, 2048), (4096, 8192)]:
    t=time.time(); A, b, c, xs = synth.planted_lp(0, m, n); print("gen", m, n, time.time()-t, flush=True)
    t=time.time(); ctx.upload_arrays(A, b, c); print("upload", time.time()-t, flush=True)
    d = np.random.default_rng(0).uniform(0.5, 2, n)
    M, ms = ctx.k_adat(d, repeats=5)
    print(f"adat {m}x{n}: {ms:.3f} ms  {m*(m+1)*n/ms/1e9:.2f} TF/s (alg)", flush=True)
    Mf = np.tril(M) + np.tril(M, -1).T
    L, info, ms = ctx.k_potrf(Mf, repeats=3); print(f"potrf {m}: {ms:.3f} ms info={info}  {m**3/3/ms/1e9:.2f} TF/s", flush=True)
    for nr in (1, 2):
        V, ms = ctx.k_chol_solve(m, np.ones((nr, m)), repeats=3); print(f"chol_solve nrhs={nr}: {ms:.3f} ms", flush=True)
        Y, ms = ctx.k_gemv_n(np.ones((nr, n)), repeats=5); print(f"gemv_n nrhs={nr}: {ms:.3f} ms  {m*n*8/ms/1e6:.1f} GB/s", flush=True)
        U, ms = ctx.k_gemv_t(np.ones((nr, m)), repeats=5); print(f"gemv_t nrhs={nr}: {ms:.3f} ms  {m*n*8/ms/1e6:.1f} GB/s", flush=True)
    ctx.set_profiling(True)
    o = lp.InteriorPoint.default().opts()
    for rep in range(2):
        t=time.time(); rc, x, fun, its, rows = ctx.solve_raw(o, want_log=True); dt=time.time()-t
        print(f"solve rc={rc} its={its} {dt*1e3:.2f} ms  {its/dt:.1f} it/s  err_vs_xstar={np.abs(x-xs).max():.2e}", ctx.phase_times(), flush=True)
    ctx.set_profiling(False)
    t=time.time(); rc, x, fun, its, rows = ctx.solve_raw(o); dt=time.time()-t
    print(f"solve(noprof) rc={rc} its={its} {dt*1e3:.2f} ms  {its/dt:.1f} it/s", flush=True)
