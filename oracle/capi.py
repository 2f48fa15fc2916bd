"""ctypes binding of the C oracle (liboracle_ipm.so).  TEST INFRASTRUCTURE ONLY (see __init__)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_ipm.so")

OK, UNCONSTRAINED, NUMERICAL_PROBLEM, INVALID_PARAMETER, INCOMPATIBLE_DIMENSIONS, INFEASIBLE, \
    UNBOUNDED, ITERATION_LIMIT = range(8)


class Opts(C.Structure):
    _fields_ = [("tol", C.c_double), ("alpha0", C.c_double), ("max_iter", C.c_uint64),
                ("ip", C.c_int32), ("solver_type", C.c_int32), ("disp", C.c_int32)]


class IterRow(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("alpha", "rho_p", "rho_d", "rho_A", "rho_g", "rho_mu", "obj")]


class Timing(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("adat", "chol", "solves", "gemv", "rest", "total")]


def build(force: bool = False) -> str:
    """Compile the C restatement with the committed Makefile (no-op when up to date)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, u64 = C.POINTER(C.c_double), C.c_uint64
        L.oracle_default_opts.argtypes = [C.POINTER(Opts)]
        L.oracle_default_opts.restype = None
        L.oracle_problem_build.argtypes = [u64, u64, dp, dp, u64, dp, dp, dp, dp, dp, dp, C.POINTER(u64)]
        L.oracle_problem_build.restype = C.c_int
        L.oracle_ipm_solve.argtypes = [u64, u64, dp, dp, dp, C.c_double, C.POINTER(Opts), dp, dp,
                                       C.POINTER(u64), C.POINTER(IterRow), C.POINTER(Timing)]
        L.oracle_ipm_solve.restype = C.c_int
        L.oracle_iteration.argtypes = [u64, u64, dp, dp, dp, C.c_int, C.c_int, C.c_double, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp]
        L.oracle_iteration.restype = C.c_int
        L.oracle_adat.argtypes = [u64, u64, dp, dp, dp]
        L.oracle_adat.restype = None
        L.oracle_cholesky.argtypes = [u64, dp]
        L.oracle_cholesky.restype = C.c_int
        L.oracle_cholesky_solve.argtypes = [u64, dp, dp, dp]
        L.oracle_cholesky_solve.restype = None
        L.oracle_gemv_n.argtypes = [u64, u64, dp, dp, dp]
        L.oracle_gemv_n.restype = None
        L.oracle_gemv_t.argtypes = [u64, u64, dp, dp, dp]
        L.oracle_gemv_t.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_opts(**kw) -> Opts:
    o = Opts()
    lib().oracle_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def problem_build(c, A_ub=None, b_ub=None, A_eq=None, b_eq=None):
    c = _f64(c)
    n = c.shape[0]
    A_ub = _f64(np.zeros((0, n)) if A_ub is None else A_ub)
    b_ub = _f64(np.zeros(0) if b_ub is None else b_ub)
    A_eq = _f64(np.zeros((0, n)) if A_eq is None else A_eq)
    b_eq = _f64(np.zeros(0) if b_eq is None else b_eq)
    if (A_ub.ndim != 2 or A_eq.ndim != 2 or A_ub.shape[1] != n or A_eq.shape[1] != n
            or A_ub.shape[0] != b_ub.shape[0] or A_eq.shape[0] != b_eq.shape[0]):
        return INCOMPATIBLE_DIMENSIONS, None, None, None, 0   # linear_program.rs:137-143
    m_ub, m_eq = A_ub.shape[0], A_eq.shape[0]
    m, ns = m_ub + m_eq, n + m_ub
    A = np.zeros((max(m, 1), ns))
    b = np.zeros(max(m, 1))
    cs = np.zeros(ns)
    nsl = C.c_uint64(0)
    rc = lib().oracle_problem_build(n, m_ub, _p(A_ub), _p(b_ub), m_eq, _p(A_eq), _p(b_eq), _p(c),
                                    _p(A), _p(b), _p(cs), C.byref(nsl))
    if rc:
        return rc, None, None, None, 0
    return OK, A[:m], b[:m], cs, int(nsl.value)


def solve(A, b, c, c0=0.0, opts: Opts | None = None, want_log=True):
    """-> dict(status, x_slack, fun, iterations, log[rows], timing{})"""
    A, b, c = _f64(A), _f64(b), _f64(c)
    m, n = A.shape
    opts = opts or default_opts()
    x = np.full(n, np.nan)
    fun = C.c_double(np.nan)
    it = C.c_uint64(0)
    nlog = int(min(opts.max_iter, 100000)) if want_log else 0
    log = (IterRow * max(nlog, 1))()
    tm = Timing()
    rc = lib().oracle_ipm_solve(m, n, _p(A), _p(b), _p(c), float(c0), C.byref(opts), _p(x),
                                C.byref(fun), C.byref(it), log if want_log else None, C.byref(tm))
    rows = []
    if want_log:
        for i in range(min(int(it.value), nlog)):
            r = log[i]
            rows.append((r.alpha, r.rho_p, r.rho_d, r.rho_A, r.rho_g, r.rho_mu, r.obj))
    return dict(status=rc, x_slack=x if rc in (OK, ITERATION_LIMIT) else None,
                fun=fun.value if rc in (OK, ITERATION_LIMIT) else None, iterations=int(it.value),
                log=rows, timing={k: getattr(tm, k) for k, _ in Timing._fields_})


def adat(A, dinv):
    A, dinv = _f64(A), _f64(dinv)
    m, n = A.shape
    M = np.empty((m, m))
    lib().oracle_adat(m, n, _p(A), _p(dinv), _p(M))
    return M


def cholesky(M):
    L = _f64(M).copy()
    rc = lib().oracle_cholesky(L.shape[0], _p(L))
    return rc, L


def cholesky_solve(L, r):
    L, r = _f64(L), _f64(r)
    v = np.empty_like(r)
    lib().oracle_cholesky_solve(L.shape[0], _p(L), _p(r), _p(v))
    return v


def gemv_n(A, w):
    A, w = _f64(A), _f64(w)
    y = np.empty(A.shape[0])
    lib().oracle_gemv_n(A.shape[0], A.shape[1], _p(A), _p(w), _p(y))
    return y


def gemv_t(A, v):
    A, v = _f64(A), _f64(v)
    u = np.empty(A.shape[1])
    lib().oracle_gemv_t(A.shape[0], A.shape[1], _p(A), _p(v), _p(u))
    return u


def iteration(A, b, c, x, y, z, tau, kappa, ip=False, alpha0=0.99995, solver_type=0):
    """One loop body of solve_normal_form from the given iterate (oracle_iteration).
    -> dict(status, x, y, z, tau, kappa, d_x, d_y, d_z, d_tau, d_kappa, alpha)"""
    A, b, c = _f64(A), _f64(b), _f64(c)
    m, n = A.shape
    x, y, z = _f64(x).copy(), _f64(y).copy(), _f64(z).copy()
    tk = np.array([float(tau), float(kappa)])
    dx, dy, dz, dtk, al = np.empty(n), np.empty(m), np.empty(n), np.empty(2), np.empty(1)
    rc = lib().oracle_iteration(m, n, _p(A), _p(b), _p(c), int(solver_type), int(bool(ip)), float(alpha0), _p(x), _p(y), _p(z),
                                _p(tk[0:1]), _p(tk[1:2]), _p(dx), _p(dy), _p(dz), _p(dtk), _p(al))
    return dict(status=rc, x=x, y=y, z=z, tau=float(tk[0]), kappa=float(tk[1]), d_x=dx, d_y=dy, d_z=dz,
                d_tau=float(dtk[0]), d_kappa=float(dtk[1]), alpha=float(al[0]))


# ---- the f32 instantiation of the same restatement (liboracle_ipm_f32.so: -DORACLE_F32 -fsingle-precision-constant) -------
_LIB32_PATH = os.path.join(_HERE, "liboracle_ipm_f32.so")


class Opts32(C.Structure):
    _fields_ = [("tol", C.c_float), ("alpha0", C.c_float), ("max_iter", C.c_uint64),
                ("ip", C.c_int32), ("solver_type", C.c_int32), ("disp", C.c_int32)]


class IterRow32(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("alpha", "rho_p", "rho_d", "rho_A", "rho_g", "rho_mu", "obj")]


_lib32 = None


def lib32():
    global _lib32
    if _lib32 is None:
        if not os.path.exists(_LIB32_PATH):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        L = C.CDLL(_LIB32_PATH)
        fp, u64 = C.POINTER(C.c_float), C.c_uint64
        L.oracle_default_opts.argtypes = [C.POINTER(Opts32)]
        L.oracle_default_opts.restype = None
        L.oracle_ipm_solve.argtypes = [u64, u64, fp, fp, fp, C.c_float, C.POINTER(Opts32), fp, fp, C.POINTER(u64),
                                       C.POINTER(IterRow32), C.c_void_p]
        L.oracle_ipm_solve.restype = C.c_int
        _lib32 = L
    return _lib32


def solve_f32(A, b, c, c0=0.0, want_log=True, **opt_kw):
    """The reference's algorithm with F = f32 (src/float.rs:42-43): -> dict(status, x_slack float32, fun, iterations, log)."""
    A = np.ascontiguousarray(A, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    c = np.ascontiguousarray(c, dtype=np.float32)
    m, n = A.shape
    o = Opts32()
    lib32().oracle_default_opts(C.byref(o))
    for k, v in opt_kw.items():
        setattr(o, k, v)
    x = np.full(n, np.nan, dtype=np.float32)
    fun, it = C.c_float(np.nan), C.c_uint64(0)
    nlog = int(min(o.max_iter, 100000)) if want_log else 0
    log = (IterRow32 * max(nlog, 1))()
    fp = lambda a_: a_.ctypes.data_as(C.POINTER(C.c_float))
    rc = lib32().oracle_ipm_solve(m, n, fp(A), fp(b), fp(c), C.c_float(c0), C.byref(o), fp(x), C.byref(fun), C.byref(it),
                                  log if want_log else None, None)
    rows = [(r.alpha, r.rho_p, r.rho_d, r.rho_A, r.rho_g, r.rho_mu, r.obj) for r in list(log)[:min(int(it.value), nlog)]] if want_log else []
    return dict(status=rc, x_slack=x if rc in (OK, ITERATION_LIMIT) else None, fun=float(fun.value) if rc in (OK, ITERATION_LIMIT) else None,
                iterations=int(it.value), log=rows)
