"""Dense fp64 MFMA rate of the chip as a function of how long the load lasts (the pure-MFMA probe kernel, 512 workgroups):
what the 78.6 TFLOP/s peak turns into under a sustained load -- the ceiling the A.D.A^T kernel should be judged against."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lp_amd
ctx = lp_amd.Context(0)
for iters in (2000, 20000, 100000, 400000, 100000, 2000):
    tf, ms = ctx.k_mfma_f64_probe(iters)
    print(f"probe iters {iters:7d}: {ms:9.3f} ms  {tf:6.2f} TFLOP/s", flush=True)
