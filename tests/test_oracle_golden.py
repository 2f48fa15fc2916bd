"""Pins the oracle (CPU restatement, oracle/) against every known-answer test the reference holds for
this path -- SURVEY.md 4 / 8(c) -- and the two transcriptions (C, numpy) against each other.
No GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KNOWN = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))["cases"]


def _arr(v):
    return None if v is None else np.array(v, dtype=float)


@pytest.mark.parametrize("case", KNOWN, ids=[c["name"] for c in KNOWN])
@pytest.mark.parametrize("solver_type", [0, 1, 2])
def test_reference_known_answers(built, case, solver_type):
    """src/lib.rs:17-52,106-113; interior_point/mod.rs:175-194,256-344 (incl. the Inverse and
    LeastSquares arms, mod.rs:275-317)."""
    from oracle import capi, oracle_np
    args = (_arr(case["c"]), _arr(case["A_ub"]), _arr(case["b_ub"]), _arr(case["A_eq"]), _arr(case["b_eq"]))
    st, A, b, c, ns = capi.problem_build(*args)
    st2, A2, b2, c2, ns2 = oracle_np.problem_build(*args)
    assert st == st2 == 0 and ns == ns2
    assert np.array_equal(A, A2) and np.array_equal(b, b2) and np.array_equal(c, c2)
    r = capi.solve(A, b, c, 0.0, capi.default_opts(solver_type=solver_type))
    r2 = oracle_np.solve(A, b, c, 0.0, oracle_np.Opts(solver_type=solver_type))
    n = len(case["c"])
    assert r["status"] == 0 and r2.status == 0
    assert np.abs(r["x_slack"][:n] - np.array(case["x"])).max() < case["eps"]
    assert np.abs(r2.x_slack[:n] - np.array(case["x"])).max() < case["eps"]
    assert r["iterations"] == r2.iterations == case["iterations"]
    assert np.abs(r["x_slack"] - r2.x_slack).max() < 1e-9


def test_example_symmetric(built):
    """examples/symmetric.rs:10-25: N = 1000, expects all-ones within 1e-10."""
    from oracle import capi
    N = 1000
    st, A, b, c, ns = capi.problem_build(-np.ones(N), 1.0 - np.eye(N), np.full(N, N - 1.0))
    assert st == 0 and A.shape == (N, 2 * N) and ns == N
    r = capi.solve(A, b, c)
    assert r["status"] == 0 and r["iterations"] == 4
    assert np.abs(r["x_slack"][:N] - 1.0).max() < 1e-10
    assert abs(r["fun"] + 1000.0) < 1e-6


def test_c1_trajectory(built):
    """alpha and indicators of the README LP, per iteration."""
    from oracle import capi
    st, A, b, c, _ = capi.problem_build([-1.0, 4.0], [[-3.0, 1.0], [1.0, 2.0]], [6.0, 4.0], [[1.0, 1.0]], [1.0])
    assert np.array_equal(A, np.array([[-3.0, 1, 1, 0], [1, 2, 0, 1], [1, 1, 0, 0]]))
    assert np.array_equal(b, [6.0, 4.0, 1.0]) and np.array_equal(c, [-1.0, 4.0, 0.0, 0.0])
    r = capi.solve(A, b, c)
    exp = np.array([[1.0, 2.11044766e-01, 8.47942880e-01, 4.54891144e-01, 2.16243541e-01, 6.61306054e+00],
                    [0.9617608, 8.13858421e-03, 3.26994820e-02, 2.15488938e-01, 8.33906616e-03, 2.55021487e-01],
                    [0.99995, 9.28046464e-07, 3.72873650e-06, 9.70318590e-06, 9.50907428e-07, 2.90802138e-05],
                    [0.99995, 4.64026146e-11, 1.86436825e-10, 4.85149285e-10, 4.75453565e-11, 1.45401083e-09]])
    got = np.array(r["log"])[:, :6]
    assert got.shape == exp.shape
    assert np.abs(got[:3] / exp[:3] - 1.0).max() < 1e-6
    assert np.abs(got[3] / exp[3] - 1.0).max() < 1e-3          # 1e-11-level residuals: rounding noise
    assert np.abs(r["x_slack"] - np.array([1.0, 0.0, 9.0, 3.0])).max() < 1e-8
    assert abs(r["fun"] + 1.0) < 1e-8


def test_exits_never_tested_by_reference(built):
    """indicators.rs:66-83, mod.rs:118-128,232-239; linear_program.rs:134-143."""
    from oracle import capi, oracle_np
    st, A, b, c, _ = capi.problem_build([1.0, 1.0], A_eq=[[1.0, 1.0]], b_eq=[-1.0])
    assert capi.solve(A, b, c)["status"] == capi.INFEASIBLE == oracle_np.solve(A, b, c).status
    st, A, b, c, _ = capi.problem_build([-1.0, 0.0], A_eq=[[1.0, -1.0]], b_eq=[0.0])
    assert capi.solve(A, b, c)["status"] == capi.UNBOUNDED == oracle_np.solve(A, b, c).status
    assert capi.problem_build([1.0])[0] == capi.UNCONSTRAINED
    assert capi.problem_build([1.0, 2.0], A_ub=[[1.0]], b_ub=[1.0])[0] == capi.INCOMPATIBLE_DIMENSIONS
    st, A, b, c, _ = capi.problem_build([-1.0, 4.0], [[-3.0, 1.0], [1.0, 2.0]], [6.0, 4.0])
    assert capi.solve(A, b, c, 0.0, capi.default_opts(alpha0=1.0))["status"] == capi.INVALID_PARAMETER
    assert capi.solve(A, b, c, 0.0, capi.default_opts(tol=0.0))["status"] == capi.INVALID_PARAMETER
    r = capi.solve(A, b, c, 0.0, capi.default_opts(max_iter=2))
    assert r["status"] == capi.ITERATION_LIMIT and r["iterations"] == 2 and r["x_slack"] is not None


@pytest.mark.parametrize("name", ["planted_64x128_s0", "planted_100x333_s1", "planted_256x512_s0",
                                  "planted_512x1024_s0"])
def test_planted_fixtures_reproduce(built, name):
    """The committed vectors are what the oracle produces today (C) and what its numpy mirror produces."""
    from lp_amd import synth
    from oracle import capi, oracle_np
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    dims, seed = name.split("_")[1], int(name.split("_s")[1])
    m, n = (int(v) for v in dims.split("x"))
    A, b, c, xstar = synth.planted_lp(seed, m, n)
    r = capi.solve(A, b, c)
    assert r["status"] == 0 and r["iterations"] == int(g["iterations"])
    assert np.abs(r["x_slack"] - g["x_slack"]).max() < 1e-9
    r2 = oracle_np.solve(A, b, c)
    assert r2.status == 0 and r2.iterations == r["iterations"]
    assert np.abs(r2.x_slack - r["x_slack"]).max() < 1e-6       # LAPACK vs unblocked Cholesky: FP-noise floor
    assert np.abs(r["x_slack"] - xstar).max() < 1e-5


def test_kernel_restatements_agree_with_numpy(built):
    from oracle import capi
    rng = np.random.default_rng(0)
    m, n = 37, 91
    A, d = rng.standard_normal((m, n)), rng.uniform(0.1, 3.0, n)
    M = capi.adat(A, d)
    assert np.abs(M - (A * d) @ A.T).max() < 1e-11
    rc, L = capi.cholesky(M)
    assert rc == 0 and np.abs(L @ L.T - M).max() < 1e-10
    r = rng.standard_normal(m)
    assert np.abs(M @ capi.cholesky_solve(L, r) - r).max() < 1e-8
    assert capi.cholesky(-np.eye(3))[0] == 1
    w, v = rng.standard_normal(n), rng.standard_normal(m)
    assert np.abs(capi.gemv_n(A, w) - A @ w).max() < 1e-12
    assert np.abs(capi.gemv_t(A, v) - A.T @ v).max() < 1e-12


def test_f32_build_of_the_oracle_on_the_reference_known_answers():
    """liboracle_ipm_f32.so -- the same restatement with every `double` a `float` and every literal single precision (what
    the reference's generic code is for F = f32, src/float.rs:42-43) -- on the reference's own known answers
    (src/lib.rs:23-51, interior_point/mod.rs:319-344) at a tolerance f32 can reach: x to f32 accuracy.  At the default 1e-8
    it cannot converge (f32 epsilon 6e-8) and ends in NumericalProblem: the behaviour lpipm_solve_f32 is held to."""
    from oracle import capi
    A = np.array([[-3.0, 1, 1, 0], [1, 2, 0, 1], [1, 1, 0, 0]])
    r = capi.solve_f32(A, [6.0, 4, 1], [-1.0, 4, 0, 0], tol=1e-4)
    assert r["status"] == 0 and r["x_slack"].dtype == np.float32
    assert np.abs(r["x_slack"][:2] - np.array([1.0, 0.0])).max() < 1e-3 and abs(r["fun"] + 1.0) < 1e-3
    Aeq = np.array([[2.0, 1, 0], [0, 2, 1], [1, 0, 2]])
    r = capi.solve_f32(Aeq, [1.0, 2, 3], [-1.0, 4, -1.2], tol=1e-4)
    assert r["status"] == 0 and np.abs(r["x_slack"] - np.array([1 / 3, 1 / 3, 4 / 3])).max() < 1e-3
    r = capi.solve_f32(A, [6.0, 4, 1], [-1.0, 4, 0, 0], max_iter=200)          # default tol 1e-8
    assert r["status"] in (capi.NUMERICAL_PROBLEM, capi.ITERATION_LIMIT)
