"""A/B/C on one box: plain launches with a host round trip per iteration; the iteration replayed as a hipGraph
(LPIPM_GRAPH=1); plain launches with the head of the next iteration enqueued before the status read (default)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, lp_amd
    from lp_amd import synth
    for (m, n) in ((512, 1024), (1024, 2048), (4096, 8192)):
        A, b, c, xs = synth.planted_lp(0, m, n)
        ctx = lp_amd.Context(0); ctx.upload_arrays(A, b, c)
        o = lp_amd.InteriorPoint.default().opts()
        for _ in range(3): rc, x, fun, it, _ = ctx.solve_raw(o)
        reps = 20 if m < 4096 else 5
        t = time.perf_counter()
        for _ in range(reps): rc, x, fun, it, _ = ctx.solve_raw(o)
        dt = (time.perf_counter() - t) / reps
        print(f"graph={os.environ.get('LPIPM_GRAPH','0')} early-head={os.environ.get('LPIPM_SPECULATE','1')} {m}x{n}: rc={rc} it={it} {dt*1e3:.3f} ms/solve {it/dt:.1f} it/s err={abs(x-xs).max():.2e}", flush=True)
        ctx.close()
else:
    for g, sp in (("0", "0"), ("1", "1"), ("0", "1")):
        env = dict(os.environ, LPIPM_EXPERIMENTAL="1", LPIPM_GRAPH=g, LPIPM_SPECULATE=sp)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
