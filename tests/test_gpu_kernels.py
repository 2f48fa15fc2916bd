"""Per-kernel parity: every HIP stage of the hot path, called through the C ABI (include/lpipm.h),
against the oracle's restatement of the reference lines it replaces.  fp64; tolerances are relative
to the natural scale of each result and written next to each check.  Reference parity at this
granularity is unpinned (the reference has no fixture for M / factor / solve; oracle_ipm.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(3, 4), (16, 16), (100, 130), (128, 256), (200, 333), (257, 700), (512, 1024)]


def _rand_problem(m, n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n))
    d = np.exp(rng.uniform(-6, 6, n))          # x/z spans many orders of magnitude late in a solve
    return A, d, rng


@pytest.mark.parametrize("m,n", SHAPES)
def test_adat_matches_oracle(ctx, m, n):
    """newton_equations.rs:54-57.  |M_gpu - M_ref| <= 1e-13 * sqrt(n) * max|M| on the lower triangle."""
    from oracle import capi as oracle
    A, d, rng = _rand_problem(m, n, 1)
    ctx.upload_arrays(A, rng.standard_normal(m), rng.standard_normal(n))
    M, _ = ctx.k_adat(d)
    Mref = oracle.adat(A, d)
    il = np.tril_indices(m)
    err = np.abs(M[il] - Mref[il]).max()
    assert err <= 1e-13 * np.sqrt(n) * np.abs(Mref).max(), err


def test_adat_asymmetric_layout(ctx):
    """A = [I 0]-like rows with distinct scales: catches a transposed C/D fragment map (row<->col)."""
    m, n = 128, 256
    A = np.zeros((m, n))
    A[np.arange(m), np.arange(m)] = 1.0
    A[np.arange(1, m), np.arange(m - 1)] += np.arange(1, m) * 0.5     # M[i, i-1] != M[i-1, i] pattern check
    d = np.arange(1, n + 1, dtype=float)
    ctx.upload_arrays(A, np.ones(m), np.ones(n))
    M, _ = ctx.k_adat(d)
    Mref = (A * d) @ A.T
    il = np.tril_indices(m)
    assert np.abs(M[il] - Mref[il]).max() <= 1e-12 * np.abs(Mref).max()


@pytest.mark.parametrize("m", [3, 16, 100, 128, 200, 257, 512, 1024])
def test_potrf_and_solve(ctx, m):
    """newton_equations.rs:129-131 and :151-169.  L L^T reproduces M to 1e-13 relative; the solve has a
    relative residual |M v - r| / (|M||v|) <= 1e-13 and agrees with the oracle's substitution to
    1e-9 relative (cond(M) ~ 1e3..1e5 here)."""
    from oracle import capi as oracle
    rng = np.random.default_rng(m)
    B = rng.standard_normal((m, 2 * m + 3))
    M = B @ B.T + 1e-3 * np.eye(m)
    L, info, _ = ctx.k_potrf(M)
    assert info == 0
    L = np.tril(L)
    assert np.abs(L @ L.T - M).max() <= 1e-13 * np.abs(M).max() * np.sqrt(m)
    rc, Lref = oracle.cholesky(M)
    assert rc == 0
    assert np.abs(L - Lref).max() <= 1e-10 * np.abs(Lref).max()
    for nrhs in (1, 2):
        R = rng.standard_normal((nrhs, m))
        V, _ = ctx.k_chol_solve(m, R)
        for q in range(nrhs):
            resid = np.abs(M @ V[q] - R[q]).max()
            assert resid <= 1e-12 * (np.abs(M).sum(axis=1).max() * np.abs(V[q]).max()), resid
            vref = oracle.cholesky_solve(Lref, R[q])
            assert np.abs(V[q] - vref).max() <= 1e-9 * np.abs(vref).max()


def test_potrf_reports_nonpositive_pivot(ctx):
    """A failed factorisation must surface (newton_equations.rs:59-63 -> NumericalProblem)."""
    m = 200
    rng = np.random.default_rng(7)
    B = rng.standard_normal((m, m + 5))
    M = B @ B.T
    M[150, 150] = -1.0
    _, info, _ = ctx.k_potrf(M)
    assert 1 <= info <= 151


def test_potrf_is_bitwise_reproducible(ctx):
    """The diagonal-block kernel hands its tiles to waves according to the SIMD each wave landed on (hardware placement
    starts at a varying SIMD): who computes a tile must not change what is computed.  30 factorisations of one matrix,
    bit for bit; the solve through the inverses it also produces likewise."""
    m = 640
    rng = np.random.default_rng(23)
    B = rng.standard_normal((m, m + 17))
    M = B @ B.T
    R = rng.standard_normal((2, m))
    L0, info, _ = ctx.k_potrf(M)
    V0, _ = ctx.k_chol_solve(m, R)
    assert info == 0
    for _ in range(30):
        L, info, _ = ctx.k_potrf(M)
        V, _ = ctx.k_chol_solve(m, R)
        assert info == 0 and np.array_equal(np.tril(L), np.tril(L0)) and np.array_equal(V, V0)


@pytest.mark.parametrize("m,force", [(2048, True), (4096, False)])
def test_potrf_lookahead_is_bit_identical(built, monkeypatch, m, force):
    """The look-ahead of the factorisation (trailing updates split: the next outer panel's columns on the chain stream, the rest
    on a CU-masked side stream behind events; default from m = 4096, LPIPM_LOOKAHEAD=1 forces it from m = 1536)
    must give the factor of the serial schedule (LPIPM_LOOKAHEAD=0) bit for bit: every element is the same k-ordered sum
    whatever the tile shapes and the streams."""
    import lp_amd
    rng = np.random.default_rng(11)
    B = rng.standard_normal((m, m + 9))
    M = B @ B.T
    monkeypatch.setenv("LPIPM_EXPERIMENTAL", "1")
    monkeypatch.setenv("LPIPM_LOOKAHEAD", "0")
    serial = lp_amd.Context(0)
    L0, info0, _ = serial.k_potrf(M)
    serial.close()
    if force:
        monkeypatch.setenv("LPIPM_LOOKAHEAD", "1")
    else:
        monkeypatch.delenv("LPIPM_LOOKAHEAD")
        monkeypatch.delenv("LPIPM_EXPERIMENTAL")
    ahead = lp_amd.Context(0)
    for _ in range(3):
        L1, info1, _ = ahead.k_potrf(M)
        assert info0 == 0 and info1 == 0
        assert np.array_equal(np.tril(L0), np.tril(L1))
    ahead.close()


@pytest.mark.parametrize("m,n", SHAPES)
def test_gemv_n_t(ctx, m, n):
    """A.w and A^T.v (feasible_point.rs:122-123, newton_equations.rs:220,223): 1e-13*sqrt(k) relative."""
    from oracle import capi as oracle
    A, _, rng = _rand_problem(m, n, 2)
    ctx.upload_arrays(A, rng.standard_normal(m), rng.standard_normal(n))
    W = rng.standard_normal((2, n))
    V = rng.standard_normal((2, m))
    for nrhs in (1, 2):
        Y, _ = ctx.k_gemv_n(W[:nrhs])
        U, _ = ctx.k_gemv_t(V[:nrhs])
        for q in range(nrhs):
            yref, uref = oracle.gemv_n(A, W[q]), oracle.gemv_t(A, V[q])
            assert np.abs(Y[q] - yref).max() <= 1e-13 * np.sqrt(n) * max(1.0, np.abs(yref).max())
            assert np.abs(U[q] - uref).max() <= 1e-13 * np.sqrt(m) * max(1.0, np.abs(uref).max())


def test_adat_linearity_full_size(ctx):
    """Size-independent property at the headline size (m=4096, n=8192): M(d1 + d2) = M(d1) + M(d2)
    and M(e_k) = a_k a_k^T, without running the oracle at full size."""
    m, n = 4096, 8192
    rng = np.random.default_rng(3)
    A = rng.standard_normal((m, n))
    ctx.upload_arrays(A, np.zeros(m), np.zeros(n))
    d1, d2 = rng.uniform(0.1, 2.0, n), rng.uniform(0.1, 2.0, n)
    M1, _ = ctx.k_adat(d1)
    M2, _ = ctx.k_adat(d2)
    M12, _ = ctx.k_adat(d1 + d2)
    il = np.tril_indices(m)
    scale = np.abs(M12[il]).max()
    assert np.abs(M12[il] - (M1[il] + M2[il])).max() <= 1e-12 * scale
    e = np.zeros(n)
    e[1234] = 1.0
    Me, _ = ctx.k_adat(e)
    ref = np.outer(A[:, 1234], A[:, 1234])
    assert np.abs(Me[il] - ref[il]).max() <= 1e-14 * np.abs(ref).max()
    # a spot block against numpy
    blk = (A[3000:3100] * d1) @ A[100:260].T
    assert np.abs(M1[3000:3100, 100:260] - blk).max() <= 1e-13 * np.sqrt(n) * np.abs(blk).max()


def test_mfma_probe_runs(ctx):
    tf, ms = ctx.k_mfma_f64_probe(2000)
    assert tf > 1.0 and ms > 0.0


@pytest.mark.parametrize("m", [3, 64, 130, 300])
def test_qr_solve(ctx, m):
    """newton_equations.rs:133-149,155-166 (QR arms): R^-1 Q^T r against numpy.linalg.solve on a symmetric,
    NOT positive definite matrix (the arms must not depend on definiteness); singular input is reported."""
    rng = np.random.default_rng(m)
    B = rng.standard_normal((m, m))
    M = B + B.T                                  # symmetric indefinite
    R = rng.standard_normal((2, m))
    V, info, _ = ctx.k_qr_solve(M, R)
    assert info == 0
    ref = np.linalg.solve(M, R.T).T
    assert np.abs(V - ref).max() <= 1e-9 * np.abs(ref).max() * max(1.0, np.linalg.cond(M) / 1e4)
    Ms = M.copy()
    Ms[:, 1] = 0.0
    Ms[1, :] = 0.0
    _, info, _ = ctx.k_qr_solve(Ms, R[:1])
    assert info != 0


@pytest.mark.parametrize("m,nrhs", [(64, 1), (100, 2), (128, 1), (700, 2), (1024, 2), (1500, 1)])
def test_symv_residual_reads_only_the_lower_triangle(ctx, m, nrhs):
    """rho = r0 - M.v of the solve's refinement step (solver.hip chol_solve_refined): symmetric M given by its lower
    triangle alone (the upper one is filled with garbage here), against numpy on the symmetric matrix."""
    rng = np.random.default_rng(m + nrhs)
    G = rng.standard_normal((m, m))
    S = G + G.T
    L = np.tril(S) + np.triu(rng.standard_normal((m, m)) * 1e6, 1)      # what the kernel sees
    V = rng.standard_normal((nrhs, m)); R0 = rng.standard_normal((nrhs, m))
    got = ctx.k_symv_residual(L, V, R0)
    ref = R0 - V @ S
    assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("m,n", [(3, 4), (100, 333), (512, 1024), (130, 5000), (1024, 2048)])
def test_gemv_dual_one_read_of_A(ctx, m, n):
    """A.w and A^T.v from one pass over A (the residual pair, residual.rs:23,25) against the oracle's two GEMVs."""
    from lp_amd import synth
    from oracle import capi as oracle
    A, b, c, _ = synth.planted_lp(1, m, n)
    ctx.upload_arrays(A, b, c)
    rng = np.random.default_rng(m * n)
    w, v = rng.standard_normal(n), rng.standard_normal(m)
    Aw, ATv, _ = ctx.k_gemv_dual(w, v)
    p = lambda a: a.ctypes.data_as(oracle.C.POINTER(oracle.C.c_double))
    rn, rt = np.empty(m), np.empty(n)
    oracle.lib().oracle_gemv_n(m, n, p(A), p(w), p(rn))
    oracle.lib().oracle_gemv_t(m, n, p(A), p(v), p(rt))
    assert np.abs(Aw - rn).max() <= 1e-12 * max(1.0, np.abs(rn).max()) * np.sqrt(n)
    assert np.abs(ATv - rt).max() <= 1e-12 * max(1.0, np.abs(rt).max()) * np.sqrt(m)


@pytest.mark.parametrize("m,n", [(100, 130), (512, 1024), (640, 3000), (1024, 4096), (1000, 5000), (2048, 9000)])
def test_adat_units_kernel_single_lp(built, monkeypatch, m, n):
    """The (tile, chunk) units kernel with its in-launch last-arriver combine (gemm_nt_units_kernel: what lockstep batches and
    the side-by-side factorisation run) forced onto a single LP (LPIPM_ADAT_UNITS=2): against numpy, bit-identical to the
    default single-LP kernel up to n = 4096 (one canonical chunking for both), and bit-reproducible over 5 launches (the
    combine is done by whichever workgroup arrives last: the sums must not depend on who that is)."""
    import lp_amd
    from lp_amd import synth
    A, b, c, _ = synth.planted_lp(3, m, n)
    d = np.random.default_rng(m + n).uniform(1e-3, 1e3, n)
    ref = (A * d) @ A.T
    il = np.tril_indices(m)
    base = lp_amd.Context(0)
    base.upload_arrays(A, b, c)
    M0, _ = base.k_adat(d)
    base.close()
    monkeypatch.setenv("LPIPM_EXPERIMENTAL", "1")
    monkeypatch.setenv("LPIPM_ADAT_UNITS", "2")
    cx = lp_amd.Context(0)
    monkeypatch.delenv("LPIPM_ADAT_UNITS")
    monkeypatch.delenv("LPIPM_EXPERIMENTAL")
    cx.upload_arrays(A, b, c)
    M1, _ = cx.k_adat(d)
    assert np.abs(M1[il] - ref[il]).max() <= 1e-12 * np.abs(ref).max()
    if n <= 4096:
        assert np.array_equal(M1[il], M0[il])
    for _ in range(5):
        M2, _ = cx.k_adat(d)
        assert np.array_equal(M2[il], M1[il])
    cx.close()
