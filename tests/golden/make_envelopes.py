"""Adds the ORACLE'S OWN ENVELOPE to the committed fixtures (round-3 tightening of the parity tests).

For every member of tests/golden/c4_members.npz (256 planted 1024x2048 LPs) and every planted_4096x8192_s*.npz the
oracle (oracle/oracle_ipm.c) is run on the LP as generated AND on the same LP with its columns permuted (4 and 2
permutations, the ones make_c4_members.py / make_golden.py use for the noise floor: a mathematically identical problem
in which only summation orders change).  Stored per member and component:

    env_dlo = x_slack - min over all those runs,   env_dhi = max over all those runs - x_slack      (both >= 0)

as float32 rounded UP with the mantissa cut to 4 bits (never narrower than the true envelope; compresses well), and
iterations_all = the iteration count of every run (run 0 = the unpermuted one).  Runs that stop at ANOTHER iteration
count are part of the envelope: on those members the oracle itself does not agree with itself about the count, and
whatever it returns under either count is "what the reference returns".

The parity tests then require of EVERY member -- the ones whose count differs and the ones whose `floor` is inf
included --   x_oracle_min - 1e-6 <= x_gpu <= x_oracle_max + 1e-6   componentwise, and the device's iteration count to
be one the oracle produced on that LP.

Run from the repo root:  python tests/golden/make_envelopes.py [--procs 6] [--only c4|c3]
"""
import argparse
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def round_up_f32(a, keep_bits=4):
    """float32 >= a (a >= 0) whose mantissa has only `keep_bits` leading bits set."""
    a = np.asarray(a, dtype=np.float64)
    f = a.astype(np.float32)
    f = np.where(f.astype(np.float64) < a, np.nextafter(f, np.float32(np.inf)), f).astype(np.float32)
    u = f.view(np.uint32)
    drop = np.uint32((1 << (23 - keep_bits)) - 1)
    u2 = np.where((u & drop) != 0, (u | drop) + np.uint32(1), u).astype(np.uint32)   # next multiple: carries into the exponent
    out = u2.view(np.float32)
    assert np.all(out.astype(np.float64) >= a)
    return out


def envelope(args):
    seed, m, n, nperm = args
    from lp_amd import synth
    from oracle import capi as oracle
    A, b, c, _ = synth.planted_lp(seed, m, n)
    r = oracle.solve(A, b, c, want_log=False)
    assert r["status"] == 0
    lo, hi, its = r["x_slack"].copy(), r["x_slack"].copy(), [r["iterations"]]
    for k in range(nperm):
        perm = np.random.default_rng(1000 * (k + 1) + seed).permutation(n)
        r2 = oracle.solve(np.ascontiguousarray(A[:, perm]), b, np.ascontiguousarray(c[perm]), want_log=False)
        assert r2["status"] == 0
        x2 = np.empty(n)
        x2[perm] = r2["x_slack"]
        lo, hi = np.minimum(lo, x2), np.maximum(hi, x2)
        its.append(r2["iterations"])
    return seed, r["x_slack"], round_up_f32(r["x_slack"] - lo), round_up_f32(hi - r["x_slack"]), np.array(its)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--only", choices=("c4", "c3"), default=None)
    a = ap.parse_args()
    if a.only in (None, "c4"):
        path = os.path.join(HERE, "c4_members.npz")
        g = dict(np.load(path))
        m, n = int(g["m"]), int(g["n"])
        with Pool(a.procs) as p:
            rows = p.map(envelope, [(int(s), m, n, 4) for s in g["seeds"]], chunksize=1)
        rows.sort(key=lambda r: r[0])
        assert all(np.array_equal(r[1], g["x_slack"][i]) for i, r in enumerate(rows)), "the oracle no longer reproduces the fixture"
        g["env_dlo"] = np.stack([r[2] for r in rows])
        g["env_dhi"] = np.stack([r[3] for r in rows])
        g["iterations_all"] = np.stack([r[4] for r in rows])
        np.savez_compressed(path, **g)
        w = (g["env_dlo"].astype(np.float64) + g["env_dhi"]).max(axis=1)
        print("c4: envelope width per member: median %.2e max %.2e; members wider than 1e-6: %d; members with more than one "
              "iteration count: %d" % (np.median(w), w.max(), int((w > 1e-6).sum()),
                                       int((g["iterations_all"].min(axis=1) != g["iterations_all"].max(axis=1)).sum())), flush=True)
    if a.only in (None, "c3"):
        with Pool(min(a.procs, 4)) as p:
            rows = p.map(envelope, [(s, 4096, 8192, 2) for s in range(4)], chunksize=1)
        for seed, x, dlo, dhi, its in rows:
            path = os.path.join(HERE, f"planted_4096x8192_s{seed}.npz")
            g = dict(np.load(path))
            assert np.array_equal(x, g["x_slack"]), "the oracle no longer reproduces the fixture"
            g["env_dlo"], g["env_dhi"], g["iterations_all"] = dlo, dhi, its
            np.savez_compressed(path, **g)
            print("c3 seed", seed, "iterations", its.tolist(), "envelope width max %.2e" % float((dlo.astype(np.float64) + dhi).max()), flush=True)


if __name__ == "__main__":
    main()
