"""Golden vectors for BASELINE config C4's real members: planted dense LPs 1024x2048, seeds 0..255
(SURVEY.md 8d generator, lp_amd/csrc/synth.cpp), solved by the C oracle (oracle/oracle_ipm.c) with the
reference's default options.

For every seed the file holds the oracle's x_slack, fun, iteration count, its distance to the planted vertex,
and the oracle's OWN rounding-noise floor on that LP: the same LP with its columns permuted (a mathematically
identical problem that only changes summation orders, SURVEY.md 8d "FP-noise floor") solved again, NPERM times,
and the largest |x - x_permuted|_inf recorded (inf if a permuted run stops at another iteration).  The GPU parity tests require the oracle's iteration count and
|x_gpu - x_oracle|_inf <= max(1e-6, 10 * floor) per member.

The reference itself cannot run here (Rust, no toolchain): these vectors come from the restatement, which the
reference's known answers pin end to end (tests/test_oracle_golden.py).

`dtau_margin` (see margin() below) marks the members whose LAST step is ill-determined in fp64 whatever the
implementation; the parity tests let those take another number of iterations than the oracle did.

Run from the repo root:  python tests/golden/make_c4_members.py [--seeds 256] [--procs 8]
                         OPENBLAS_NUM_THREADS=1 python tests/golden/make_c4_members.py --margins-only
"""
import argparse
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

M, N = 1024, 2048
NPERM = 4      # permuted re-solves per member; the floor is the largest |dx| among them


def one(seed):
    from lp_amd import synth
    from oracle import capi as oracle
    A, b, c, xstar = synth.planted_lp(seed, M, N)
    r = oracle.solve(A, b, c, want_log=True)
    floor, its2 = 0.0, []
    for k in range(NPERM):        # the same LP with its columns permuted: only summation orders change
        perm = np.random.default_rng(1000 * (k + 1) + seed).permutation(N)
        r2 = oracle.solve(np.ascontiguousarray(A[:, perm]), b, np.ascontiguousarray(c[perm]), want_log=False)
        x2 = np.empty(N)
        x2[perm] = r2["x_slack"]
        its2.append(r2["iterations"])
        floor = max(floor, float(np.abs(x2 - r["x_slack"]).max()) if r2["iterations"] == r["iterations"] else float("inf"))
    alphas = np.array([row[0] for row in r["log"]])
    return (seed, r["status"], r["x_slack"], r["fun"], r["iterations"], float(np.abs(r["x_slack"] - xstar).max()),
            floor, max(its2, key=lambda v: abs(v - r["iterations"])), alphas)


def margin(seed):
    """How well determined the LAST iteration's direction is: Delta::compute (delta.rs:29-32) divides by
    kappa/tau + (-c.p + b.q), and near the optimum c.p and b.q agree to many digits.  margin = |that denominator| /
    max(|c.p|, |b.q|) at the oracle's last iterate (numpy transcription of the oracle, traced).  A Cholesky solve of
    these normal equations (cond ~1e9) delivers q to ~1e-8 relative in ANY fp64 implementation, so below ~1e-6 the
    denominator -- hence d_tau, hence the whole last step -- has one or two significant digits: whether that step is
    clean (alpha = 0.99995) or poor (and the solver needs more iterations) is decided by rounding."""
    import scipy.linalg as sla
    from lp_amd import synth
    from oracle import oracle_np
    A, b, c, _ = synth.planted_lp(seed, M, N)
    tr = []
    oracle_np.solve(A, b, c, trace=tr)
    x, y, z, tau, kappa = tr[-1]
    d = x / z
    cf = sla.cho_factor(A @ (d[:, None] * A.T), lower=True)
    q = sla.cho_solve(cf, b + A @ (d * c))
    p = d * (A.T @ q - c)
    cp, bq = c @ p, b @ q
    return abs(kappa / tau - cp + bq) / max(abs(cp), abs(bq))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=256)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--margins-only", action="store_true", help="add dtau_margin to the existing file")
    a = ap.parse_args()
    if a.margins_only:
        os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
        path = os.path.join(HERE, "c4_members.npz")
        g = dict(np.load(path))
        with Pool(a.procs) as p:
            g["dtau_margin"] = np.array(p.map(margin, [int(s) for s in g["seeds"]], chunksize=1))
        np.savez_compressed(path, **g)
        print("margins: min %.1e median %.1e; below 2e-6: %d" % (g["dtau_margin"].min(), np.median(g["dtau_margin"]),
                                                                   int((g["dtau_margin"] < 2e-6).sum())))
        return
    with Pool(a.procs) as p:
        rows = p.map(one, range(a.seeds), chunksize=1)
    rows.sort(key=lambda r: r[0])
    assert all(r[1] == 0 for r in rows), [r[0] for r in rows if r[1] != 0]
    maxit = max(r[4] for r in rows)
    alpha = np.full((len(rows), maxit), np.nan)
    for i, r in enumerate(rows):
        alpha[i, :len(r[8])] = r[8]
    np.savez_compressed(os.path.join(HERE, "c4_members.npz"), m=M, n=N,
                        seeds=np.array([r[0] for r in rows]), x_slack=np.stack([r[2] for r in rows]),
                        fun=np.array([r[3] for r in rows]), iterations=np.array([r[4] for r in rows]),
                        xstar_err=np.array([r[5] for r in rows]), floor=np.array([r[6] for r in rows]),
                        iterations_permuted=np.array([r[7] for r in rows]), alpha=alpha)
    fl = np.array([r[6] for r in rows])
    print("members", len(rows), "iterations", np.bincount([r[4] for r in rows]), "floor median %.2e max %.2e; > 1e-7: %d"
          % (np.median(fl), fl.max(), int((fl > 1e-7).sum())))


if __name__ == "__main__":
    main()
