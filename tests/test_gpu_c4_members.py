"""BASELINE config C4's real members (planted dense LPs 1024x2048, seeds 0..255) against the committed oracle
vectors (tests/golden/c4_members.npz, generator beside it), through BOTH device paths:

  (a) lpipm_solve           -- one LP at a time;
  (b) lpipm_solve_lockstep  -- chunks of 32 members advancing together (what a C4 shard runs).

Checked per member: status Optimal, the oracle's iteration count, and |x_gpu - x_oracle|_inf <= max(1e-6, 10 * floor)
where `floor` is the oracle's OWN rounding noise on that LP (largest |dx| among four column-permuted re-solves of the
same LP with the oracle, recorded in the fixture).  1e-6 is BASELINE.json's tolerance; it holds for the members whose
floor is below 1e-7 -- for the others (listed by the test output) no two correct fp64 solvers agree to 1e-6: the oracle
does not agree with itself.

The two paths must agree with each other BIT FOR BIT: an LP goes through the same kernels with the same summation
orders alone and as a batch member (canonical chunked summation in A.D.A^T, kernels_gemm.hip; super-block width a
function of m alone, solver.hip)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "c4_members.npz")
CHUNK = 32


@pytest.fixture(scope="module")
def c4_runs(built):
    import lp_amd
    from lp_amd import synth
    g = np.load(GOLD)
    m, n, K = int(g["m"]), int(g["n"]), len(g["seeds"])
    o = lp_amd.InteriorPoint.default().opts()
    one, lock = lp_amd.Context(0), lp_amd.Context(0)
    xs = {"single": np.empty((K, n)), "lock": np.empty((K, n))}
    its = {"single": np.zeros(K, int), "lock": np.zeros(K, int)}
    for s0 in range(0, K, CHUNK):
        seeds = [int(s) for s in g["seeds"][s0:s0 + CHUNK]]
        probs = [synth.planted_lp(s, m, n) for s in seeds]
        for k, (A, b, c, _) in enumerate(probs):
            one.upload_arrays(A, b, c)
            rc, x, fun, it, _ = one.solve_raw(o)
            assert rc == 0, (seeds[k], rc)
            xs["single"][s0 + k], its["single"][s0 + k] = x, it
        lock.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
        for k, (st, x, fun, it) in enumerate(lock.solve_lockstep(o)):
            assert st == 0, (seeds[k], st)
            xs["lock"][s0 + k], its["lock"][s0 + k] = x, it
    one.close()
    lock.close()
    return g, xs, its


def test_lockstep_is_bit_identical_to_single(c4_runs):
    g, xs, its = c4_runs
    assert np.array_equal(its["single"], its["lock"])
    diff = np.abs(xs["single"] - xs["lock"]).max(axis=1)
    assert np.array_equal(xs["single"], xs["lock"]), f"members that differ: {np.where(diff > 0)[0].tolist()}, worst {diff.max():.3e}"


@pytest.mark.parametrize("path", ["single", "lock"])
def test_c4_members_match_oracle(c4_runs, path):
    g, xs, its = c4_runs
    floor = g["floor"]
    bar = np.maximum(1e-6, 10.0 * floor)
    err = np.abs(xs[path] - g["x_slack"]).max(axis=1)
    # a member whose oracle count changes when only the oracle's summation orders change (floor = inf) may take either
    wrong_it = np.where((its[path] != g["iterations"]) & ~((its[path] == g["iterations_permuted"]) & ~np.isfinite(floor)))[0]
    over = np.where(err > bar)[0]
    loose = np.where(floor > 1e-7)[0]
    print(f"\n[{path}] |x - x_oracle|: median {np.median(err):.2e}, max {err.max():.2e}; members > 1e-6: "
          f"{np.where(err > 1e-6)[0].tolist()}; members whose oracle floor exceeds 1e-7 (bar = 10 x floor): "
          + ", ".join(f"{int(s)}:{floor[s]:.1e}" for s in loose))
    assert len(wrong_it) == 0, [(int(s), int(its[path][s]), int(g["iterations"][s])) for s in wrong_it]
    assert len(over) == 0, [(int(s), float(err[s]), float(bar[s])) for s in over]
