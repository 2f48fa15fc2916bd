"""Runs the A.D.A^T kernel of the C3 LP back to back for > 2 s on random data and prints the in-kernel clock
(needs the diagnostic build of adat_clock_patch.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, lp_amd
from lp_amd import synth
A, b, c, xs = synth.planted_lp(0, 4096, 8192)
ctx = lp_amd.Context(0); ctx.upload_arrays(A, b, c)
d = np.random.default_rng(0).uniform(0.5, 2.0, 8192)
for reps in (50, 1000):
    M, ms = ctx.k_adat(d, repeats=reps)
    print(f"{reps} launches back to back: {ms:.4f} ms per launch = {4096*4097*8192/ms/1e9:.2f} TFLOP/s", flush=True)
