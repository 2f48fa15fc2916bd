"""C4 shard (32 x 1024x2048) as ONE lockstep batch of 32 vs TWO lockstep batches of 16 solved concurrently from two host
threads (own context, own stream each): does the hardware overlap one half's A.D.A^T with the other half's chain?"""
import os, sys, time
os.environ["LPIPM_EXPERIMENTAL"] = "1"      # the library reads its measurement knobs only with the master switch on
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth
K, m, n, reps = 32, 1024, 2048, 5
probs = [synth.planted_lp(s, m, n) for s in range(K)]
o = lp.InteriorPoint.default().opts()

def make(lo, hi):
    cx = lp.Context(0)
    cx.upload_lockstep([p[0] for p in probs[lo:hi]], [p[1] for p in probs[lo:hi]], [p[2] for p in probs[lo:hi]])
    cx.solve_lockstep(o)
    return cx

def run(ctxs, stagger=0.0):
    def work(cx, delay):
        if delay: time.sleep(delay)
        for _ in range(reps):
            cx.solve_lockstep(o)
    ths = [threading.Thread(target=work, args=(cx, i * stagger)) for i, cx in enumerate(ctxs)]
    t = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    return time.perf_counter() - t

one = make(0, 32)
dt = run([one]); print(f"one batch of 32: {K*reps/dt:.1f} LP/s", flush=True)
one.close()
a, b = make(0, 16), make(16, 32)
dt = run([a]); print(f"one batch of 16 alone: {16*reps/dt:.1f} LP/s", flush=True)
for stag in (0.0, 0.0015):
    dt = run([a, b], stag); print(f"two batches of 16, two threads, stagger {stag*1e3:.1f} ms: {K*reps/(dt - stag):.1f} LP/s", flush=True)
a.close(); b.close()
q = [make(8 * i, 8 * i + 8) for i in range(4)]
dt = run(q); print(f"four batches of 8, four threads: {K*reps/dt:.1f} LP/s", flush=True)
