"""C4-shaped shard on one GPU: K independent 1024x2048 LPs (BASELINE config 4 has 32 per GPU).
  (a) lpipm_solve_batch, one-by-one path at several concurrencies (upload of each LP inside the timed region)
  (b) lpipm_solve_batch with same-shape members grouped into lockstep batches (upload inside the timed region)
  (c) lockstep batch with the inputs already resident in HBM (lpipm_upload_lockstep untimed, lpipm_solve_lockstep timed)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import _capi, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m_ = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n_ = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
probs = [synth.planted_lp(s, m_, n_) for s in range(K)]
dp = C.POINTER(C.c_double)
As = [p[0] for p in probs]; bs = [p[1] for p in probs]; cs = [p[2] for p in probs]
xs = [np.full(n_, np.nan) for _ in range(K)]
arr = lambda lst: (dp * K)(*[a.ctypes.data_as(dp) for a in lst])
m = (C.c_uint64 * K)(*([m_] * K)); n = (C.c_uint64 * K)(*([n_] * K))
fun = (C.c_double * K)(); its = (C.c_uint64 * K)(); st = (C.c_int32 * K)()
o = lp.InteriorPoint.default().opts()
ctx = lp.default_context(0)
L = _capi.lib()
def report(tag, dt, rc=0):
    err = max(np.abs(xs[i] - probs[i][3]).max() for i in range(K))
    print(f"{tag:44s} rc={rc} {K} LPs in {dt*1e3:8.2f} ms  {sum(its)/dt:8.1f} it/s  {K/dt:7.1f} LP/s  ok={all(s == 0 for s in st)} max_err_vs_xstar={err:.2e}", flush=True)
L.lpipm_set_batch_lockstep(ctx._h, 0)
for conc in (1, 8):
    L.lpipm_set_batch_concurrency(ctx._h, conc)
    for rep in range(4):
        t = time.perf_counter()
        rc = L.lpipm_solve_batch(ctx._h, K, m, n, arr(As), arr(bs), arr(cs), None, C.byref(o), arr(xs), fun, its, st)
        dt = time.perf_counter() - t
    report(f"(a) one by one, concurrency {conc}", dt, rc)
for grp in (-1, 8, 16, 32):
    L.lpipm_set_batch_lockstep(ctx._h, grp)
    for rep in range(4):
        t = time.perf_counter()
        rc = L.lpipm_solve_batch(ctx._h, K, m, n, arr(As), arr(bs), arr(cs), None, C.byref(o), arr(xs), fun, its, st)
        dt = time.perf_counter() - t
    report(f"(b) lockstep chunks of {grp if grp > 0 else 'auto'}, upload timed, pipelined", dt, rc)
t = time.perf_counter()
ctx.upload_lockstep(As, bs, cs)
tu = time.perf_counter() - t
for rep in range(3):
    t = time.perf_counter()
    rc = L.lpipm_solve_lockstep(ctx._h, C.byref(o), arr(xs), fun, its, st)
    dt = time.perf_counter() - t
report(f"(c) lockstep, resident inputs (upload {tu*1e3:.1f} ms)", dt, rc)
