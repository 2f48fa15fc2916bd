"""numpy mirror of oracle_ipm.c -- the same restatement of the reference algorithm, on NumPy/OpenBLAS.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never from lp_amd/.  Two uses:
  * cross-checks the C restatement (two independent transcriptions of the same reference lines);
  * "B-strong" CPU baseline: stand-in for the reference's `openblas-system`
    feature (Cargo.toml:23-24) -- same algorithm, LAPACK Cholesky, all host cores.
Every function cites the reference lines it follows (paths relative to
/root/reference/src/solvers/interior_point/ unless stated).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

OK, UNCONSTRAINED, NUMERICAL_PROBLEM, INVALID_PARAMETER, INCOMPATIBLE_DIMENSIONS, INFEASIBLE, \
    UNBOUNDED, ITERATION_LIMIT = range(8)


@dataclass
class Opts:  # mod.rs:41-60
    tol: float = 1e-8
    alpha0: float = 0.99995
    max_iter: int = 1000
    ip: bool = True
    solver_type: int = 0  # 0 Cholesky, 1 Inverse, 2 LeastSquares
    disp: bool = False


@dataclass
class Result:
    status: int
    x_slack: np.ndarray | None
    fun: float | None
    iterations: int
    log: list = field(default_factory=list)      # rows (alpha, rho_p, rho_d, rho_A, rho_g, rho_mu, obj)
    timing: dict = field(default_factory=dict)


def problem_build(c, A_ub=None, b_ub=None, A_eq=None, b_eq=None):
    """src/linear_program.rs:125-169 -> (status, A, b, c_slack, n_slack)."""
    c = np.asarray(c, dtype=np.float64)
    n = c.shape[0]
    A_ub = np.zeros((0, n)) if A_ub is None else np.asarray(A_ub, dtype=np.float64)
    b_ub = np.zeros(0) if b_ub is None else np.asarray(b_ub, dtype=np.float64)
    A_eq = np.zeros((0, n)) if A_eq is None else np.asarray(A_eq, dtype=np.float64)
    b_eq = np.zeros(0) if b_eq is None else np.asarray(b_eq, dtype=np.float64)
    if A_ub.shape[0] + A_eq.shape[0] == 0:                                   # :134-136
        return UNCONSTRAINED, None, None, None, 0
    if (A_ub.ndim != 2 or A_eq.ndim != 2 or A_ub.shape[1] != A_eq.shape[1] or A_eq.shape[1] != n
            or A_ub.shape[0] != b_ub.shape[0] or A_eq.shape[0] != b_eq.shape[0]):  # :137-143
        return INCOMPATIBLE_DIMENSIONS, None, None, None, 0
    m_ub, m_eq = A_ub.shape[0], A_eq.shape[0]
    A1 = np.concatenate([A_ub, A_eq], axis=0)                                # :145
    A2 = np.concatenate([np.eye(m_ub), np.zeros((m_eq, m_ub))], axis=0)      # :147-154
    A = np.ascontiguousarray(np.concatenate([A1, A2], axis=1))               # :155
    b = np.concatenate([b_ub, b_eq])                                         # :157
    cs = np.concatenate([c, np.zeros(m_ub)])                                 # :159
    return OK, A, b, cs, m_ub                                                # :161


def _residuals(A, b, c, x, y, z, tau, kappa):
    """residual.rs:13-44."""
    rho_p = np.sqrt(np.sum((b * tau - A @ x) ** 2))
    rho_d = np.sqrt(np.sum((c * tau - A.T @ y - z) ** 2))
    rho_g = abs(kappa + c @ x - b @ y)
    rho_mu = (x @ z + tau * kappa) / (x.shape[0] + 1)
    return rho_p, rho_d, rho_g, rho_mu


def _indicators(A, b, c, c0, x, y, z, tau, kappa, r0):
    """indicators.rs:37-55."""
    obj = c @ (x / tau) + c0
    bty = b @ y
    rho_A = abs(c @ x - bty) / (tau + abs(b @ y))
    rp, rd, rg, rmu = _residuals(A, b, c, x, y, z, tau, kappa)
    return dict(rho_p=rp / max(r0[0], 1.0), rho_d=rd / max(r0[1], 1.0), rho_A=rho_A,
                rho_g=rg / max(r0[2], 1.0), rho_mu=rmu / r0[3], obj=obj, bty=bty)


def _status(ind, tau, kappa, tol):
    """indicators.rs:66-83."""
    tau_too_small = tau < tol * max(kappa, 1.0)
    inf1 = (ind["rho_p"] < tol and ind["rho_d"] < tol and ind["rho_g"] < tol) and tau_too_small
    inf2 = ind["rho_mu"] < tol and tau_too_small
    if inf1 or inf2:
        return INFEASIBLE if ind["bty"] > tol else UNBOUNDED
    if ind["rho_p"] < tol and ind["rho_d"] < tol and ind["rho_A"] < tol:
        return OK
    return -1


class _EqSolver:
    """newton_equations.rs:48-64, :129-169 (Cholesky via LAPACK, Inverse/LeastSquares via QR)."""

    def __init__(self, A, x, z, solver_type, timing):
        import scipy.linalg as sla
        self.sla = sla
        self.timing = timing
        self.Dinv = x / z                                                    # :54
        t0 = time.perf_counter()
        self.M = A @ (self.Dinv[:, None] * A.T)                              # :55-57
        timing["adat"] += time.perf_counter() - t0
        self.kind = None
        self.ok = self._factor(solver_type)

    def _factor(self, kind):
        t0 = time.perf_counter()
        self.kind = kind
        try:
            if kind == 0:
                self.factor = self.sla.cho_factor(self.M, lower=True, check_finite=False)
            else:
                self.factor = np.linalg.qr(self.M)
            ok = True
        except (np.linalg.LinAlgError, ValueError):
            ok = False
        self.timing["chol"] += time.perf_counter() - t0
        return ok

    def solve(self, r):                                                      # :151-169
        t0 = time.perf_counter()
        if self.kind == 0:
            v = self.sla.cho_solve(self.factor, r, check_finite=False)
        else:
            Q, R = self.factor
            v = self.sla.solve_triangular(R, Q.T @ r, lower=False, check_finite=False)
        self.timing["solves"] += time.perf_counter() - t0
        return v

    def sym_solve(self, A, r1, r2):                                          # :214-225
        t0 = time.perf_counter()
        r = r2 + A @ (self.Dinv * r1)
        self.timing["gemv"] += time.perf_counter() - t0
        v = self.solve(r)
        t0 = time.perf_counter()
        u = self.Dinv * (A.T @ v - r1)
        self.timing["gemv"] += time.perf_counter() - t0
        return u, v


def _delta(A, b, c, x, z, tau, kappa, rhat, S):
    """delta.rs:21-49 + newton_equations.rs:176-210 (NaN check; no fallback needed with LAPACK)."""
    p, q = S.sym_solve(A, c, b)
    u, v = S.sym_solve(A, rhat["d"] - rhat["xs"] / x, rhat["p"])
    if np.isnan(p).any() or np.isnan(q).any():
        return None
    d_tau = (rhat["g"] + 1.0 / tau * rhat["tk"] - (-(c @ u) + b @ v)) / \
            (1.0 / tau * kappa + (-(c @ p) + b @ q))
    d_x = u + p * d_tau
    d_y = v + q * d_tau
    d_z = (rhat["xs"] - z * d_x) / x
    d_kappa = 1.0 / tau * (rhat["tk"] - kappa * d_tau)
    return dict(d_x=d_x, d_y=d_y, d_z=d_z, d_tau=d_tau, d_kappa=d_kappa)


def _step_size(x, z, tau, kappa, d, alpha0):
    """feasible_point.rs:53-72."""
    def fold(dv, v):
        neg = dv < 0
        return min(1.0, np.min(v[neg] / -dv[neg])) if neg.any() else 1.0
    ax, az = fold(d["d_x"], x), fold(d["d_z"], z)
    at = min(1.0, tau / -d["d_tau"]) if d["d_tau"] < 0 else 1.0
    ak = min(1.0, kappa / -d["d_kappa"]) if d["d_kappa"] < 0 else 1.0
    return min(1.0, ax, at, az, ak) * alpha0


def solve(A, b, c, c0=0.0, opts: Opts | None = None, trace: list | None = None) -> Result:
    """mod.rs:199-240 + :161-168 on the slack-form problem.  trace (tests / diagnostics): receives a copy of the
    iterate (x, y, z, tau, kappa) at the start of every iteration."""
    opts = opts or Opts()
    if not (0.0 < opts.alpha0 < 1.0) or not (opts.tol > 0.0):                 # mod.rs:118-128
        return Result(INVALID_PARAMETER, None, None, 0)
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    m, n = A.shape
    timing = dict(adat=0.0, chol=0.0, solves=0.0, gemv=0.0, rest=0.0, total=0.0)
    t_start = time.perf_counter()
    x, y, z, tau, kappa = np.ones(n), np.zeros(m), np.ones(n), 1.0, 1.0       # feasible_point.rs:24-39
    r0 = _residuals(A, b, c, x, y, z, tau, kappa)
    ip = bool(opts.ip)
    log = []
    status, it = ITERATION_LIMIT, 0
    for it in range(1, opts.max_iter + 1):                                    # mod.rs:213
        if trace is not None:
            trace.append((x.copy(), y.copy(), z.copy(), tau, kappa))
        # get_delta, feasible_point.rs:110-152
        gamma = 1.0 if ip else 0.0
        eta = 1.0 if ip else 1.0 - gamma
        t0 = time.perf_counter()
        r_P = b * tau - A @ x
        r_D = c * tau - A.T @ y - z
        timing["gemv"] += time.perf_counter() - t0
        r_G = c @ x - b @ y + kappa
        mu = (x @ z + tau * kappa) / (n + 1)
        S = _EqSolver(A, x, z, opts.solver_type, timing)
        if not S.ok:
            status = NUMERICAL_PROBLEM
            break
        rhat = dict(p=r_P * eta, d=r_D * eta, g=r_G * eta, xs=(x * -1.0) * z + gamma * mu,
                    tk=gamma * mu - tau * kappa)                              # rhat.rs:17-35
        pred = _delta(A, b, c, x, z, tau, kappa, rhat, S)
        if pred is None:
            status = NUMERICAL_PROBLEM
            break
        alpha = _step_size(x, z, tau, kappa, pred, 1.0)                       # :134
        gamma = 10.0 if ip else (1.0 - alpha) ** 2 * min(0.1, 1.0 - alpha)    # :156-165
        eta = 1.0 if ip else 1.0 - gamma
        if ip:                                                                # rhat.rs:51-60
            a2 = alpha * alpha
            xs = (x * -1.0) * z - (pred["d_x"] * pred["d_z"]) * a2 + (1.0 - alpha) * gamma * mu
            tk = (1.0 - alpha) * gamma * mu - tau * kappa - a2 * pred["d_tau"] * pred["d_kappa"]
        else:                                                                 # rhat.rs:62-66
            xs = (x * -1.0) * z + gamma * mu - (pred["d_x"] * pred["d_z"])
            tk = gamma * mu - tau * kappa - pred["d_tau"] * pred["d_kappa"]
        rhat = dict(p=r_P * eta, d=r_D * eta, g=r_G * eta, xs=xs, tk=tk)
        d = _delta(A, b, c, x, z, tau, kappa, rhat, S)
        if d is None:
            status = NUMERICAL_PROBLEM
            break
        alpha = 1.0 if ip else _step_size(x, z, tau, kappa, d, opts.alpha0)   # mod.rs:216-221
        x = x + d["d_x"] * alpha                                              # feasible_point.rs:76-106
        y = y + d["d_y"] * alpha
        z = z + d["d_z"] * alpha
        tau = tau + d["d_tau"] * alpha
        kappa = kappa + d["d_kappa"] * alpha
        if ip:
            x, z, tau, kappa = np.maximum(x, 1.0), np.maximum(z, 1.0), max(tau, 1.0), max(kappa, 1.0)
        ip = False
        ind = _indicators(A, b, c, c0, x, y, z, tau, kappa, r0)               # mod.rs:225
        log.append((alpha, ind["rho_p"], ind["rho_d"], ind["rho_A"], ind["rho_g"], ind["rho_mu"],
                    ind["obj"]))
        st = _status(ind, tau, kappa, opts.tol)
        if st >= 0:
            status = st
            break
    timing["total"] = time.perf_counter() - t_start
    if status in (OK, ITERATION_LIMIT):
        xs_out = x / tau
        return Result(status, xs_out, float(c @ xs_out + c0), it, log, timing)
    return Result(status, None, None, it, log, timing)
