"""A.D.A^T kernel variants and the side-by-side factorisation, on the GPU box:
   python scripts/adat_variants.py            -> per-launch ms of k_adat (units kernel vs round-2 kernel, bit comparison),
                                                 solve it/s at C3 with LPIPM_OVERLAP on/off, C4 lockstep LP/s both kernels."""
import os, sys, time
os.environ["LPIPM_EXPERIMENTAL"] = "1"      # the library reads its measurement knobs only with the master switch on
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth

def ctx_with(**env):
    for k, v in env.items():
        os.environ[k] = str(v)
    c = lp.Context(0)
    for k in env:
        del os.environ[k]
    return c

def adat(m, n, reps=10):
    A, b, c, _ = synth.planted_lp(0, m, n)
    d = np.random.default_rng(1).uniform(0.1, 3.0, n)
    out = {}
    for name, env in (("units", {"LPIPM_ADAT_UNITS": 2}), ("round2", {"LPIPM_ADAT_UNITS": 0})):
        cx = ctx_with(LPIPM_OVERLAP=0, **env)
        cx.upload_arrays(A, b, c)
        cx.k_adat(d, 2)
        M, ms = cx.k_adat(d, reps)
        out[name] = (M, ms)
        cx.close()
    il = np.tril_indices(m)
    same = np.array_equal(out["units"][0][il], out["round2"][0][il])
    diff = np.abs(out["units"][0][il] - out["round2"][0][il]).max()
    fl = m * (m + 1.0) * n
    print(f"adat {m}x{n}: units {out['units'][1]:.3f} ms ({fl/out['units'][1]/1e9:.1f} TF), round2 {out['round2'][1]:.3f} ms "
          f"({fl/out['round2'][1]/1e9:.1f} TF); bit-identical {same}, max |d| {diff:.2e}", flush=True)

def solve(m, n, reps, **env):
    A, b, c, xs = synth.planted_lp(0, m, n)
    cx = ctx_with(**env)
    cx.upload_arrays(A, b, c)
    o = lp.InteriorPoint.default().opts()
    rc, x, fun, it, _ = cx.solve_raw(o)
    t = time.perf_counter()
    for _ in range(reps):
        rc, x, fun, it, _ = cx.solve_raw(o)
    dt = (time.perf_counter() - t) / reps
    cx.set_profiling(1)
    cx.solve_raw(o)
    pt = cx.phase_times()
    cx.close()
    print(f"solve {m}x{n} {env}: rc {rc} it {it} {it/dt:.1f} it/s ({dt*1e3/it:.3f} ms/it) err {np.abs(x-xs).max():.2e} "
          f"adat {pt['adat_ms']/it:.3f} potrf {pt['potrf_ms']/it:.3f} trsv {pt['trsv_ms']/it:.3f} gemv {pt['gemv_ms']/it:.3f} vec {pt['vec_ms']/it:.3f}", flush=True)
    return x

def lockstep(K, m, n, reps, **env):
    probs = [synth.planted_lp(s, m, n) for s in range(K)]
    cx = ctx_with(**env)
    cx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
    o = lp.InteriorPoint.default().opts()
    res = cx.solve_lockstep(o)
    t = time.perf_counter()
    for _ in range(reps):
        res = cx.solve_lockstep(o)
    dt = (time.perf_counter() - t) / reps
    cx.set_profiling(1)
    cx.solve_lockstep(o)
    pt = cx.phase_times()
    its = max(r[3] for r in res)
    cx.close()
    print(f"lockstep {K}x({m}x{n}) {env}: {K/dt:.1f} LP/s, {dt*1e3/its:.3f} ms per lockstep iteration; adat {pt['adat_ms']/its:.3f} "
          f"potrf {pt['potrf_ms']/its:.3f} trsv {pt['trsv_ms']/its:.3f} gemv {pt['gemv_ms']/its:.3f} vec {pt['vec_ms']/its:.3f}", flush=True)
    return np.stack([r[1] for r in res])

what = sys.argv[1:] or ["adat", "c3", "c4", "c2"]
if "adat" in what:
    for (m, n) in ((512, 1024), (1024, 2048), (2048, 4096), (4096, 8192), (1000, 5000), (3000, 3500), (2048, 16384), (6144, 12288)):
        adat(m, n)
if "c3" in what:
    x0 = solve(4096, 8192, 5, LPIPM_OVERLAP=0, LPIPM_ADAT_UNITS=0)
    x1 = solve(4096, 8192, 5, LPIPM_OVERLAP=0)
    x2 = solve(4096, 8192, 5, LPIPM_OVERLAP=1)
    x3 = solve(4096, 8192, 5, LPIPM_OVERLAP=1, LPIPM_OVERLAP_CUS=2)
    print("c3: |x_units - x_round2|", np.abs(x1 - x0).max(), " |x_overlap - x_units|", np.abs(x2 - x1).max(), np.abs(x3 - x1).max(), flush=True)
if "c4" in what:
    a = lockstep(32, 1024, 2048, 5, LPIPM_ADAT_UNITS=0)
    b = lockstep(32, 1024, 2048, 5)
    print("c4: units bit-identical to round2:", np.array_equal(a, b), flush=True)
if "c4only" in what:
    lockstep(32, 1024, 2048, 5)
if "c2" in what:
    solve(512, 1024, 50, LPIPM_OVERLAP=0, LPIPM_ADAT_UNITS=0)
    solve(512, 1024, 50, LPIPM_OVERLAP=0)
    solve(2048, 4096, 10, LPIPM_OVERLAP=0)
    solve(2048, 4096, 10, LPIPM_OVERLAP=1)
