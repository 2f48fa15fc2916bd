"""Host -> HBM upload rate of one LP through lpipm_upload (pageable numpy source)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lp_amd
from lp_amd import synth
ctx = lp_amd.Context(0)
for (m, n) in ((1024, 2048), (4096, 8192), (8192, 16384)):
    A, b, c, _ = synth.planted_lp(0, m, n)
    ctx.upload_arrays(A, b, c)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); ctx.upload_arrays(A, b, c); ts.append(time.perf_counter() - t)
    print(f"{m}x{n}: {A.nbytes/2**20:.0f} MiB in {min(ts)*1e3:.2f} ms = {A.nbytes/min(ts)/1e9:.1f} GB/s (re-upload, same geometry)", flush=True)
