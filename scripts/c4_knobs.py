"""C4 shard (32 x 1024x2048 lockstep) under the environment given on the command line; prints LP/s and the phase split."""
import os, sys, time
os.environ["LPIPM_EXPERIMENTAL"] = "1"      # the library reads its measurement knobs only with the master switch on
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth
K, m, n, reps = 32, 1024, 2048, 5
probs = [synth.planted_lp(s, m, n) for s in range(K)]
o = lp.InteriorPoint.default().opts()
ref = None
for spec in sys.argv[1:]:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    os.environ.update(env)
    cx = lp.Context(0)
    cx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
    res = cx.solve_lockstep(o)
    t = time.perf_counter()
    for _ in range(reps):
        res = cx.solve_lockstep(o)
    dt = (time.perf_counter() - t) / reps
    cx.set_profiling(1); cx.solve_lockstep(o); pt = cx.phase_times(); cx.set_profiling(0)
    its = max(r[3] for r in res)
    X = np.stack([r[1] for r in res])
    if ref is None: ref = X
    print(f"{spec or '(default)':45s} {K/dt:7.1f} LP/s  {dt*1e3/its:.3f} ms/it  adat {pt['adat_ms']/its:.3f} potrf {pt['potrf_ms']/its:.3f} "
          f"trsv {pt['trsv_ms']/its:.3f} gemv {pt['gemv_ms']/its:.3f} vec {pt['vec_ms']/its:.3f}  same bits as first: {np.array_equal(X, ref)}", flush=True)
    cx.close()
    for k in env: del os.environ[k]
