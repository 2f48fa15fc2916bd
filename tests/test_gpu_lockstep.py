"""Lockstep batches (BASELINE config C4: a shard of independent same-shape LPs on one GPU): every kernel launch
covers all LPs of the batch.  Each LP must come out exactly as if it had been solved alone: same status, same
iteration count, |dx| <= 1e-6 against the oracle -- including LPs that stop at different iterations, infeasible
and unbounded members, and a member that hits the iteration limit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lps(shapes_seeds):
    from lp_amd import synth
    return [synth.planted_lp(s, m, n)[:3] for (m, n, s) in shapes_seeds]


@pytest.mark.parametrize("m,n,count", [(96, 200, 5), (256, 512, 8), (130, 333, 3), (64, 150, 18)])
def test_lockstep_matches_oracle_and_single(ctx, m, n, count):
    import lp_amd
    from oracle import capi as oracle
    probs = _lps([(m, n, s) for s in range(count)])
    o = lp_amd.InteriorPoint.default().opts()
    ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
    res = ctx.solve_lockstep(o)
    res2 = ctx.solve_lockstep(o)                                   # resident batch solved again: deterministic
    for i, (A, b, c) in enumerate(probs):
        ref = oracle.solve(A, b, c)
        st, x, fun, it = res[i]
        assert st == ref["status"] == 0 and it == ref["iterations"]
        assert np.abs(x - ref["x_slack"]).max() <= 1e-6
        assert abs(fun - ref["fun"]) <= 1e-6 * max(1.0, abs(ref["fun"]))
        assert np.array_equal(x, res2[i][1]) and it == res2[i][3]
    # and against the single-LP path of the same library
    single = lp_amd.Context(0)
    for i, (A, b, c) in enumerate(probs):
        single.upload_arrays(A, b, c)
        rc, x1, f1, it1, _ = single.solve_raw(o)
        # bit for bit: the same kernels with the same arguments, alone or as a member (count 18: two half-batch views on
        # two host threads; n <= 1024: the fused single-workgroup vector stage)
        assert rc == 0 and it1 == res[i][3] and np.array_equal(x1, res[i][1])
    single.close()


def test_lockstep_members_stop_at_different_iterations(ctx):
    """One infeasible, one unbounded, one optimal-with-c0 and planted members of the same shape: finished members
    are frozen while the others iterate on."""
    import lp_amd
    from lp_amd import _capi, synth
    from oracle import capi as oracle
    m, n = 2, 4
    rng = np.random.default_rng(7)
    feas = []
    for s in range(3):
        A, b, c, _ = synth.planted_lp(s, m, n)
        feas.append((A, b, c))
    inf = (np.array([[1.0, 1.0, 1.0, 1.0], [1.0, 0.0, 1.0, 0.0]]), np.array([-1.0, 1.0]), np.ones(4))
    unb = (np.array([[1.0, -1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 1.0]]), np.array([0.0, 1.0]), np.array([-1.0, 0.0, 0.0, 0.0]))
    probs = [feas[0], inf, feas[1], unb, feas[2]]
    c0s = [0.0, 0.0, 2.5, 0.0, -1.0]
    o = lp_amd.InteriorPoint.default().opts()
    ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs], c0s)
    res = ctx.solve_lockstep(o)
    its = []
    for i, (A, b, c) in enumerate(probs):
        ref = oracle.solve(A, b, c, c0s[i])
        st, x, fun, it = res[i]
        assert st == ref["status"], (i, st, ref["status"])
        assert it == ref["iterations"], (i, it, ref["iterations"])
        its.append(it)
        if st == 0:
            assert np.abs(x - ref["x_slack"]).max() <= 1e-6 and abs(fun - ref["fun"]) <= 1e-6
        else:
            assert x is None
    assert res[1][0] == _capi.INFEASIBLE and res[3][0] == _capi.UNBOUNDED
    assert len(set(its)) > 1                                       # the point of the test


def test_lockstep_iteration_limit_and_invalid_options(ctx):
    import lp_amd
    from lp_amd import _capi
    from oracle import capi as oracle
    probs = _lps([(64, 128, s) for s in range(3)])
    ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
    o = lp_amd.InteriorPoint.custom().max_iter(2).build().opts()
    res = ctx.solve_lockstep(o)
    for i, (A, b, c) in enumerate(probs):
        ref = oracle.solve(A, b, c, 0.0, oracle.default_opts(max_iter=2))
        assert res[i][0] == ref["status"] == _capi.ITERATION_LIMIT and res[i][3] == 2
        assert np.abs(res[i][1] - ref["x_slack"]).max() <= 1e-6   # payload x / tau (mod.rs:237-239)
    bad = lp_amd.InteriorPoint.default().opts()
    bad.alpha0 = 1.5
    with pytest.raises(lp_amd.InvalidParameter):
        ctx.solve_lockstep(bad)
    qr = lp_amd.InteriorPoint.custom().solver_type(lp_amd.EquationSolverType.Inverse).build().opts()
    with pytest.raises(lp_amd.BackendError):
        ctx.solve_lockstep(qr)
    # the context goes back to a single LP without residue
    A, b, c = probs[0]
    ctx.upload_arrays(A, b, c)
    rc, x, fun, it, _ = ctx.solve_raw(lp_amd.InteriorPoint.default().opts())
    assert rc == 0 and np.abs(x - oracle.solve(A, b, c)["x_slack"]).max() <= 1e-6


def test_solve_batch_groups_same_shapes(built):
    """lpipm_solve_batch with mixed shapes: equal shapes go through lockstep groups, odd ones one by one; results
    are the same with grouping switched off."""
    import lp_amd
    from lp_amd import _capi
    from oracle import capi as oracle
    shapes = [(64, 128, 0), (64, 128, 1), (40, 100, 2), (64, 128, 3), (100, 260, 4), (100, 260, 5), (64, 128, 6)]
    probs = _lps(shapes)
    k = len(probs)
    dp = C.POINTER(C.c_double)
    arr = lambda lst: (dp * k)(*[a.ctypes.data_as(dp) for a in lst])
    m = (C.c_uint64 * k)(*[p[0].shape[0] for p in probs]); n = (C.c_uint64 * k)(*[p[0].shape[1] for p in probs])
    o = lp_amd.InteriorPoint.default().opts()
    ctx = lp_amd.Context(0)
    out = {}
    for mode in (-1, 0, 3):
        xs = [np.full(p[0].shape[1], np.nan) for p in probs]
        fun = (C.c_double * k)(); its = (C.c_uint64 * k)(); st = (C.c_int32 * k)()
        assert _capi.lib().lpipm_set_batch_lockstep(ctx._h, mode) == 0
        rc = _capi.lib().lpipm_solve_batch(ctx._h, k, m, n, arr([p[0] for p in probs]), arr([p[1] for p in probs]),
                                           arr([p[2] for p in probs]), None, C.byref(o), arr(xs), fun, its, st)
        assert rc == 0
        out[mode] = (xs, list(fun), list(its), list(st))
    ctx.close()
    for i, (A, b, c) in enumerate(probs):
        ref = oracle.solve(A, b, c)
        for mode in out:
            xs, fun, its, st = out[mode]
            assert st[i] == 0 and its[i] == ref["iterations"], (mode, i)
            assert np.abs(xs[i] - ref["x_slack"]).max() <= 1e-6
            assert abs(fun[i] - ref["fun"]) <= 1e-6 * max(1.0, abs(ref["fun"]))


def test_lockstep_c4_shape_properties(built):
    """BASELINE config 4's member shape (1024x2048), 8 members in lockstep: size-independent properties of every
    answer (primal feasibility, x >= 0, objective equal to the planted optimum's) and agreement with the planted
    vertex and with the one-at-a-time path.  (The oracle needs ~10 s per such LP: not used here.)"""
    import lp_amd
    from lp_amd import synth
    probs = [synth.planted_lp(s, 1024, 2048) for s in range(8)]     # seeds with a rounding noise floor < 1e-6
    o = lp_amd.InteriorPoint.default().opts()
    ctx = lp_amd.Context(0)
    ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])
    res = ctx.solve_lockstep(o)
    single = lp_amd.Context(0)
    for i, (A, b, c, xstar) in enumerate(probs):
        st, x, fun, it = res[i]
        assert st == 0
        assert np.abs(A @ x - b).max() <= 1e-6 * max(1.0, np.abs(b).max())
        assert x.min() >= -1e-12
        assert abs(fun - c @ xstar) <= 1e-6 * max(1.0, abs(c @ xstar))
        assert np.abs(x - xstar).max() <= 1e-6
        single.upload_arrays(A, b, c)
        rc, x1, f1, it1, _ = single.solve_raw(o)
        assert rc == 0 and it1 == it and np.abs(x1 - x).max() <= 1e-6
    single.close()
    ctx.close()
