"""Regenerates the golden fixtures under tests/golden/.

known_answers.json : the reference's OWN known-answer tests for this path, transcribed as data
                     (inputs + expected x + epsilon), with the file:line each comes from.  These pin
                     the oracle (tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_solve.py).
planted_*.npz      : planted dense LPs (lp_amd.synth, seed in the name) solved by the C oracle
                     (oracle/oracle_ipm.c): x_slack, fun, iterations and the per-iteration log.
                     The reference itself cannot run here (Rust, no toolchain), so these vectors come
                     from the restatement, which the known answers above pin end to end.
Run from the repo root:  python tests/golden/make_golden.py [--big]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

KNOWN = [
    dict(name="readme_lp", source="src/lib.rs:23-51,106-113; interior_point/mod.rs:256-273",
         c=[-1.0, 4.0], A_ub=[[-3.0, 1.0], [1.0, 2.0]], b_ub=[6.0, 4.0], A_eq=[[1.0, 1.0]], b_eq=[1.0],
         x=[1.0, 0.0], eps=1e-6, iterations=4),
    dict(name="custom_doctest_ub_only", source="interior_point/mod.rs:175-194",
         c=[-1.0, 4.0], A_ub=[[-3.0, 1.0], [1.0, 2.0]], b_ub=[6.0, 4.0], A_eq=None, b_eq=None,
         x=[4.0, 0.0], eps=1e-6, iterations=5),
    dict(name="linprog_eq_only", source="interior_point/mod.rs:319-331",
         c=[-1.0, 4.0, -1.2], A_ub=None, b_ub=None,
         A_eq=[[2.0, 1.0, 0.0], [0.0, 2.0, 1.0], [1.0, 0.0, 2.0]], b_eq=[1.0, 2.0, 3.0],
         x=[1.0 / 3.0, 1.0 / 3.0, 4.0 / 3.0], eps=1e-6, iterations=3),
    dict(name="linprog_ub_only", source="interior_point/mod.rs:332-344",
         c=[-1.0, 4.0, -1.2], A_ub=[[2.0, 1.0, 0.0], [0.0, 2.0, 1.0], [1.0, 0.0, 2.0]], b_ub=[1.0, 2.0, 3.0],
         A_eq=None, b_eq=None, x=[0.5, 0.0, 1.25], eps=1e-6, iterations=6),
]
# examples/symmetric.rs:10-25 is generated, not stored: A_ub = 1 - I (N = 1000), b_ub = N - 1, c = -1,
# expected x = 1 within 1e-10 (tests build it on the fly).

PLANTED = [(0, 64, 128), (1, 100, 333), (0, 256, 512), (0, 512, 1024)]
PLANTED_BIG = [(0, 4096, 8192), (1, 4096, 8192), (2, 4096, 8192), (3, 4096, 8192)]


def main():
    from lp_amd import synth
    from oracle import capi as oracle
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(dict(note="iterations = count observed with the oracle (SURVEY.md 4 lists the same)",
                       cases=KNOWN), f, indent=1)
    todo = PLANTED + (PLANTED_BIG if "--big" in sys.argv else [])
    if "--only-big" in sys.argv:     # python tests/golden/make_golden.py --only-big <seed>   (one process per seed)
        todo = [t for t in PLANTED_BIG if t[0] == int(sys.argv[sys.argv.index("--only-big") + 1])]
    if "--only-small" in sys.argv:
        todo = PLANTED
    for seed, m, n in todo:
        A, b, c, xstar = synth.planted_lp(seed, m, n)
        r = oracle.solve(A, b, c)
        assert r["status"] == 0
        # the oracle's own rounding noise on this LP: the same LP with its columns permuted, solved again (2 times);
        # floor = largest |dx| among them, inf if a permuted run stops at another iteration (its count is recorded)
        floor, its_perm = 0.0, []
        for k in range(2):
            perm = np.random.default_rng(1000 * (k + 1) + seed).permutation(n)
            r2 = oracle.solve(np.ascontiguousarray(A[:, perm]), b, np.ascontiguousarray(c[perm]), want_log=False)
            x2 = np.empty(n)
            x2[perm] = r2["x_slack"]
            its_perm.append(r2["iterations"])
            floor = max(floor, float(np.abs(x2 - r["x_slack"]).max()) if r2["iterations"] == r["iterations"] else float("inf"))
        np.savez_compressed(os.path.join(HERE, f"planted_{m}x{n}_s{seed}.npz"), x_slack=r["x_slack"],
                            fun=r["fun"], iterations=r["iterations"], log=np.array(r["log"]),
                            xstar_err=np.abs(r["x_slack"] - xstar).max(), floor=floor, iterations_permuted=np.array(its_perm))
        print(m, n, seed, "iterations", r["iterations"], its_perm, "max|x - x*|", np.abs(r["x_slack"] - xstar).max(), "floor", floor)


if __name__ == "__main__":
    main()
