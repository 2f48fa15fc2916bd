"""lp_amd -- MI355X-native interior-point LP hot path behind the reference's own API.

Host-side mirror (Python, over the C ABI of include/lpipm.h) of the public interface of
sebasv/lp (crate `ripped` 0.1.1) for the one path this package accelerates:

    reference (Rust)                                   here
    -----------------------------------------------    ------------------------------------------
    Problem::target(&c).ub(&A,&b).eq(&A,&b).build()    Problem.target(c).ub(A, b).eq(A, b).build()
    InteriorPoint::default()                           InteriorPoint.default()
    InteriorPoint::custom().tol(..)...build()          InteriorPoint.custom().tol(..)...build()
    solver.solve(&problem) -> Result<OptimizeResult>   solver.solve(problem) -> OptimizeResult | raises
    res.x() / res.fun() / res.iteration()              res.x() / res.fun() / res.iteration()
    LinearProgramError::{Unconstrained, ...}           LinearProgramError subclasses of the same names

(src/linear_program.rs:24-170, src/solvers/mod.rs:12-49, src/solvers/interior_point/mod.rs:41-197,
src/error.rs:7-29.)  A Rust `Result::Err(e)` becomes a raised exception of the matching class.
All numerics run in liblpipm.so on the GPU; importing this package never imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import enum

import numpy as np

from . import _capi

__all__ = ["Problem", "ProblemBuilder", "InteriorPoint", "InteriorPointBuilder", "EquationSolverType",
           "OptimizeResult", "Solver", "Context", "LinearProgramError", "Unconstrained",
           "NumericalProblem", "InvalidParameter", "IncompatibleInputDimensions", "Infeasible",
           "Unbounded", "IterationLimitExceeded", "BackendError"]


# ------------------------------------------------------------------------------ error.rs:7-29
class LinearProgramError(Exception):
    """error.rs:10-28.  `code` is the lpipm_status the C ABI returned."""
    code = -1

    def __init__(self, msg: str | None = None):
        super().__init__(msg if msg is not None else _capi.strerror(self.code))


class Unconstrained(LinearProgramError):
    code = _capi.UNCONSTRAINED


class NumericalProblem(LinearProgramError):
    code = _capi.NUMERICAL_PROBLEM


class InvalidParameter(LinearProgramError):
    code = _capi.INVALID_PARAMETER

    def __init__(self, what: str = ""):
        super().__init__(f"A parameter was set to an invalid value: {what}")


class IncompatibleInputDimensions(LinearProgramError):
    code = _capi.INCOMPATIBLE_DIMENSIONS


class Infeasible(LinearProgramError):
    code = _capi.INFEASIBLE


class Unbounded(LinearProgramError):
    code = _capi.UNBOUNDED


class IterationLimitExceeded(LinearProgramError):
    """error.rs:26-28: carries the best x / tau after the final iteration (mod.rs:237-239)."""
    code = _capi.ITERATION_LIMIT

    def __init__(self, x: np.ndarray):
        super().__init__()
        self.x = x


class BackendError(RuntimeError):
    """HIP/driver failure (status >= 100): no analogue in the reference; never silently ignored."""


_BY_CODE = {c.code: c for c in (Unconstrained, NumericalProblem, IncompatibleInputDimensions, Infeasible, Unbounded)}


def _raise_for(code: int, x=None):
    if code == _capi.OK:
        return
    if code == _capi.ITERATION_LIMIT:
        raise IterationLimitExceeded(x)
    if code == _capi.INVALID_PARAMETER:
        raise InvalidParameter("rejected by the backend")
    if code in _BY_CODE:
        raise _BY_CODE[code]()
    raise BackendError(f"lpipm status {code}: {_capi.strerror(code)} {_capi.last_error_detail()}")


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ------------------------------------------------------------------------------ linear_program.rs
class Problem:
    """A linear program in slack form (linear_program.rs:24-30): min c'x st A x == b, x >= 0."""

    def __init__(self, A, b, c, c0, n_slack, parts=None, dtype=np.float64):
        self._A, self._b, self._c, self._c0, self._n_slack = A, b, c, float(c0), int(n_slack)
        # F of Problem<F> (linear_program.rs:24): float32 when every array the builder was given is float32 (the reference
        # infers F from its ndarray arguments), float64 otherwise.  A Problem<f32> is solved by InteriorPoint<f32>
        # (lpipm_solve_f32: every operation in f32, src/float.rs:42-43).
        self.dtype = np.dtype(dtype)
        # the `ub` / `eq` blocks the builder was given: (A_ub, b_ub, A_eq, b_eq, c).  With them the slack-form matrix
        # is assembled on the device (lpipm_upload_ub_eq) and the host copy below exists only if A() is asked for.
        self._parts = parts

    @staticmethod
    def target(c) -> "ProblemBuilder":          # linear_program.rs:37-39
        return ProblemBuilder(c)

    def A(self) -> np.ndarray:                   # :42-44
        if self._A is None:                      # lazily: [[A_ub, I], [A_eq, 0]] (:145-156)
            A_ub, _, A_eq, _, c = self._parts
            m_ub, m_eq, n = A_ub.shape[0], A_eq.shape[0], c.shape[0]
            A = np.zeros((m_ub + m_eq, n + m_ub), dtype=self.dtype)
            A[:m_ub, :n] = A_ub
            A[m_ub:, :n] = A_eq
            A[np.arange(m_ub), n + np.arange(m_ub)] = 1.0
            self._A = A
        return self._A

    def b(self) -> np.ndarray:                   # :47-49
        return self._b

    def c(self) -> np.ndarray:                   # :52-54
        return self._c

    def c0(self) -> float:                       # :57-59
        return self._c0

    def n_slack(self) -> int:
        return self._n_slack

    def denormalize_x_into(self, x_slack: np.ndarray) -> np.ndarray:   # :65-69
        return np.array(x_slack[: x_slack.shape[0] - self._n_slack])


class ProblemBuilder:
    """linear_program.rs:72-170."""

    def __init__(self, c):
        self._c = c
        self._ub = None
        self._eq = None

    def ub(self, A, b) -> "ProblemBuilder":      # :93-96
        self._ub = (A, b)
        return self

    def eq(self, A, b) -> "ProblemBuilder":      # :102-105
        self._eq = (A, b)
        return self

    def build(self) -> Problem:                  # :125-169
        given = [self._c] + [a for pair in (self._ub, self._eq) if pair is not None for a in pair]
        f32 = all(isinstance(a, np.ndarray) and a.dtype == np.float32 for a in given)
        c = _f64(self._c)
        if c.ndim != 1:
            raise IncompatibleInputDimensions()
        n = c.shape[0]
        A_ub, b_ub = self._ub if self._ub is not None else (np.zeros((0, n)), np.zeros(0))
        A_eq, b_eq = self._eq if self._eq is not None else (np.zeros((0, n)), np.zeros(0))
        A_ub, b_ub, A_eq, b_eq = _f64(A_ub), _f64(b_ub), _f64(A_eq), _f64(b_eq)
        if A_ub.ndim != 2 or A_eq.ndim != 2 or b_ub.ndim != 1 or b_eq.ndim != 1:
            raise IncompatibleInputDimensions()
        m_ub, m_eq = A_ub.shape[0], A_eq.shape[0]
        if m_ub + m_eq == 0:                                                 # :134-136
            raise Unconstrained()
        if (A_ub.shape[1] != A_eq.shape[1] or A_eq.shape[1] != n or m_ub != b_ub.shape[0]
                or m_eq != b_eq.shape[0]):                                   # :137-143
            raise IncompatibleInputDimensions()
        b = np.concatenate([b_ub, b_eq])                                     # :157-158
        cs = np.concatenate([c, np.zeros(m_ub)])                             # :159-160
        if f32:     # Problem<f32>: the f32 values as given (exactly representable in the f64 working copies above)
            t = np.float32
            return Problem(None, b.astype(t), cs.astype(t), 0.0, m_ub,
                           parts=(A_ub.astype(t), b_ub.astype(t), A_eq.astype(t), b_eq.astype(t), c.astype(t)), dtype=t)
        return Problem(None, b, cs, 0.0, m_ub, parts=(A_ub, b_ub, A_eq, b_eq, c))   # :161; A on demand


# ------------------------------------------------------------------------------ solvers/mod.rs
class OptimizeResult:
    """solvers/mod.rs:19-49."""

    def __init__(self, x, fun, iteration):
        self._x, self._fun, self._iteration = x, float(fun), int(iteration)

    def iteration(self) -> int:
        return self._iteration

    def fun(self) -> float:
        return self._fun

    def x(self) -> np.ndarray:
        return self._x


class Solver:
    """solvers/mod.rs:12-16."""

    def solve(self, problem: Problem) -> OptimizeResult:
        raise NotImplementedError


class EquationSolverType(enum.IntEnum):
    """newton_equations.rs:37-46."""
    Cholesky = 0
    Inverse = 1
    LeastSquares = 2


# ------------------------------------------------------------------------------ device context
class Context:
    """One lpipm_ctx: device buffers + stream of one (thread, device).  `InteriorPoint.solve` keeps one
    per device; bench/tests use it directly to separate the one-time upload from the timed solve."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = _capi.lib().lpipm_create(int(device), C.byref(self._h))
        if rc != _capi.OK:
            raise BackendError(f"lpipm_create(device={device}) failed: {_capi.strerror(rc)}: "
                               f"{_capi.last_error_detail()} -- a HIP device is required, there is no CPU path")
        self.device = int(device)
        self.n = self.m = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _capi.lib().lpipm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, problem: Problem, use_slack_structure: bool = True):
        if use_slack_structure and getattr(problem, "_parts", None) is not None:
            # device-side assembly: the ub / eq blocks as given, no host slack matrix (lpipm_upload_ub_eq)
            A_ub, b_ub, A_eq, b_eq, c = problem._parts
            m_ub, m_eq, n = A_ub.shape[0], A_eq.shape[0], c.shape[0]
            rc = _capi.lib().lpipm_upload_ub_eq(self._h, n, m_ub, _p(A_ub) if m_ub else None, n, _p(b_ub) if m_ub else None,
                                                m_eq, _p(A_eq) if m_eq else None, n, _p(b_eq) if m_eq else None,
                                                _p(c), float(problem.c0()))
            _raise_for(rc)
            self.m, self.n = m_ub + m_eq, n + m_ub
            return self
        A, b, c = _f64(problem.A()), _f64(problem.b()), _f64(problem.c())
        return self.upload_arrays(A, b, c, problem.c0(), problem.n_slack() if use_slack_structure else 0)

    def upload_arrays(self, A, b, c, c0=0.0, n_slack=0):
        A, b, c = _f64(A), _f64(b), _f64(c)
        if A.ndim != 2 or b.shape != (A.shape[0],) or c.shape != (A.shape[1],):
            raise IncompatibleInputDimensions()
        m, n = A.shape
        rc = _capi.lib().lpipm_upload_slack(self._h, m, n, _p(A), n, _p(b), _p(c), float(c0), int(n_slack))
        _raise_for(rc)
        self.m, self.n = m, n
        return self

    def set_collective(self, rank: int, world: int, collective):
        """`collective.cfn` is an lpipm_allreduce_fn thunk (lp_amd.colsplit.TorchCollective); kept alive here."""
        self._collective = collective
        cfn = C.cast(collective.cfn, C.c_void_p) if collective is not None else None
        _raise_for(_capi.lib().lpipm_set_collective(self._h, int(rank), int(world), cfn, None))
        _raise_for(_capi.lib().lpipm_set_collective_on_stream(self._h, int(bool(getattr(collective, "on_stream", False)))))
        return self

    def upload_column_block(self, A_local, b, c_local, n_total: int, c0=0.0):
        """This rank's columns of one LP split over ranks (lpipm_upload_nsplit); solve_raw then returns
        this rank's slice of x."""
        A, b, c = _f64(A_local), _f64(b), _f64(c_local)
        if A.ndim != 2 or b.shape != (A.shape[0],) or c.shape != (A.shape[1],):
            raise IncompatibleInputDimensions()
        m, nl = A.shape
        _raise_for(_capi.lib().lpipm_upload_nsplit(self._h, m, int(n_total), nl, _p(A), nl, _p(b), _p(c), float(c0)))
        self.m, self.n = m, nl
        return self

    def upload_lockstep(self, As, bs, cs, c0s=None):
        """`len(As)` LPs of one shape resident at once (lpipm_upload_lockstep); solve with solve_lockstep."""
        As = [_f64(A) for A in As]; bs = [_f64(b) for b in bs]; cs = [_f64(c) for c in cs]
        K = len(As)
        if K < 1 or len(bs) != K or len(cs) != K:
            raise IncompatibleInputDimensions()
        m, n = As[0].shape
        for A, b, c in zip(As, bs, cs):
            if A.shape != (m, n) or b.shape != (m,) or c.shape != (n,):
                raise IncompatibleInputDimensions()
        dp = C.POINTER(C.c_double)
        arr = lambda lst: (dp * K)(*[_p(a) for a in lst])
        c0 = (C.c_double * K)(*[float(v) for v in c0s]) if c0s is not None else None
        _raise_for(_capi.lib().lpipm_upload_lockstep(self._h, K, m, n, arr(As), arr(bs), arr(cs), c0))
        self._lock = (K, m, n, As, bs, cs)      # keep the host arrays alive only for the duration of the call chain
        self.m, self.n = m, n
        return self

    def solve_lockstep(self, opts: "_capi.Opts"):
        """-> list of (status, x_slack | None, fun, iterations), one per LP of the last upload_lockstep"""
        K, m, n = self._lock[:3]
        dp = C.POINTER(C.c_double)
        xs = [np.full(n, np.nan) for _ in range(K)]
        xp = (dp * K)(*[_p(x) for x in xs])
        fun = (C.c_double * K)(); its = (C.c_uint64 * K)(); st = (C.c_int32 * K)()
        rc = _capi.lib().lpipm_solve_lockstep(self._h, C.byref(opts), xp, fun, its, st)
        if rc != _capi.OK:
            _raise_for(rc)
        out = []
        for i in range(K):
            has_x = st[i] in (_capi.OK, _capi.ITERATION_LIMIT)
            out.append((int(st[i]), xs[i] if has_x else None, fun[i] if has_x else None, int(its[i])))
        return out

    def solve_lockstep_device(self, opts: "_capi.Opts", x_dev_ptr: int, row_stride: int):
        """Like solve_lockstep, the solutions left in HBM: x / tau of LP i goes to the device row
        x_dev_ptr + i * row_stride doubles (lpipm_solve_lockstep_device).  -> list of (status, fun | None, iterations)"""
        K = self._lock[0]
        fun = (C.c_double * K)(); its = (C.c_uint64 * K)(); st = (C.c_int32 * K)()
        rc = _capi.lib().lpipm_solve_lockstep_device(self._h, C.byref(opts), C.c_void_p(int(x_dev_ptr)), int(row_stride),
                                                     fun, its, st)
        if rc != _capi.OK:
            _raise_for(rc)
        return [(int(st[i]), fun[i] if st[i] in (_capi.OK, _capi.ITERATION_LIMIT) else None, int(its[i])) for i in range(K)]

    def solve_batch_device(self, problems, opts: "_capi.Opts", x_dev_ptr: int, row_stride: int):
        """lpipm_solve_batch_device over [(A, b, c, c0), ...]: member i's x / tau goes to the device row
        x_dev_ptr + i * row_stride doubles.  -> list of (status, fun | None, iterations)"""
        K = len(problems)
        if K == 0:
            return []
        As = [_f64(p[0]) for p in problems]; bs = [_f64(p[1]) for p in problems]; cs = [_f64(p[2]) for p in problems]
        for A, b, c in zip(As, bs, cs):
            if A.ndim != 2 or b.shape != (A.shape[0],) or c.shape != (A.shape[1],):
                raise IncompatibleInputDimensions()
        dp = C.POINTER(C.c_double)
        arr = lambda lst: (dp * K)(*[_p(a) for a in lst])
        m = (C.c_uint64 * K)(*[A.shape[0] for A in As]); n = (C.c_uint64 * K)(*[A.shape[1] for A in As])
        c0 = (C.c_double * K)(*[float(p[3]) if len(p) > 3 else 0.0 for p in problems])
        fun = (C.c_double * K)(); its = (C.c_uint64 * K)(); st = (C.c_int32 * K)()
        rc = _capi.lib().lpipm_solve_batch_device(self._h, K, m, n, arr(As), arr(bs), arr(cs), c0, C.byref(opts),
                                                  C.c_void_p(int(x_dev_ptr)), int(row_stride), fun, its, st)
        if rc != _capi.OK:
            _raise_for(rc)
        return [(int(st[i]), fun[i] if st[i] in (_capi.OK, _capi.ITERATION_LIMIT) else None, int(its[i])) for i in range(K)]

    def solve_batch(self, problems, opts: "_capi.Opts"):
        """lpipm_solve_batch over [(A, b, c, c0), ...] (any mix of shapes; equal shapes run as lockstep batches).
        -> list of (status, x_slack | None, fun | None, iterations)"""
        K = len(problems)
        if K == 0:
            return []
        As = [_f64(p[0]) for p in problems]; bs = [_f64(p[1]) for p in problems]; cs = [_f64(p[2]) for p in problems]
        for A, b, c in zip(As, bs, cs):
            if A.ndim != 2 or b.shape != (A.shape[0],) or c.shape != (A.shape[1],):
                raise IncompatibleInputDimensions()
        dp = C.POINTER(C.c_double)
        arr = lambda lst: (dp * K)(*[_p(a) for a in lst])
        xs = [np.full(A.shape[1], np.nan) for A in As]
        m = (C.c_uint64 * K)(*[A.shape[0] for A in As]); n = (C.c_uint64 * K)(*[A.shape[1] for A in As])
        c0 = (C.c_double * K)(*[float(p[3]) if len(p) > 3 else 0.0 for p in problems])
        fun = (C.c_double * K)(); its = (C.c_uint64 * K)(); st = (C.c_int32 * K)()
        rc = _capi.lib().lpipm_solve_batch(self._h, K, m, n, arr(As), arr(bs), arr(cs), c0, C.byref(opts), arr(xs), fun, its, st)
        if rc != _capi.OK:
            _raise_for(rc)
        out = []
        for i in range(K):
            has_x = st[i] in (_capi.OK, _capi.ITERATION_LIMIT)
            out.append((int(st[i]), xs[i] if has_x else None, fun[i] if has_x else None, int(its[i])))
        return out

    def solve_raw(self, opts: "_capi.Opts", want_log: bool = False, x_dev_ptr: int | None = None):
        """-> (status, x_slack | None, fun, iterations, log rows)"""
        x = None if x_dev_ptr is not None else np.full(self.n, np.nan)
        fun, it = C.c_double(np.nan), C.c_uint64(0)
        if want_log and opts.max_iter > (1 << 20):      # the library writes one row per iteration, up to max_iter of them
            raise InvalidParameter("want_log with max_iter > 2^20")
        nlog = int(opts.max_iter) if want_log else 0
        log = (_capi.IterRow * max(nlog, 1))() if want_log else None
        if x_dev_ptr is not None:
            rc = _capi.lib().lpipm_solve_device(self._h, C.byref(opts), C.c_void_p(int(x_dev_ptr)), C.byref(fun),
                                                C.byref(it), log)
        else:
            rc = _capi.lib().lpipm_solve(self._h, C.byref(opts), _p(x), C.byref(fun), C.byref(it), log)
        rows = []
        if want_log:
            for i in range(min(int(it.value), nlog)):
                r = log[i]
                rows.append((r.alpha, r.rho_p, r.rho_d, r.rho_A, r.rho_g, r.rho_mu, r.obj))
        return rc, x, fun.value, int(it.value), rows

    def solve_f32(self, A, b, c, c0: float = 0.0, opts: "_capi.Opts | None" = None, want_log: bool = False):
        """InteriorPoint<f32>::solve on a slack-form problem (src/float.rs:42-43; lpipm_solve_f32): every operation in f32.
        -> (status, x_slack float32 | None, fun, iterations, log rows).  The context's uploaded fp64 problem, if any, is untouched."""
        A = np.ascontiguousarray(A, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
        c = np.ascontiguousarray(c, dtype=np.float32)
        if A.ndim != 2 or b.shape != (A.shape[0],) or c.shape != (A.shape[1],):
            raise IncompatibleInputDimensions()
        opts = opts or InteriorPoint.default().opts()
        m, n = A.shape
        x = np.full(n, np.nan, dtype=np.float32)
        fun, it = C.c_float(np.nan), C.c_uint64(0)
        nlog = int(opts.max_iter) if want_log else 0
        log = (C.c_float * (7 * max(nlog, 1)))() if want_log else None
        fp = lambda a_: a_.ctypes.data_as(C.POINTER(C.c_float))
        rc = _capi.lib().lpipm_solve_f32(self._h, m, n, fp(A), n, fp(b), fp(c), C.c_float(c0), C.byref(opts), fp(x), C.byref(fun),
                                         C.byref(it), C.cast(log, C.c_void_p) if want_log else None)
        rows = [tuple(float(log[7 * i + k]) for k in range(7)) for i in range(min(int(it.value), nlog))] if want_log else []
        has_x = rc in (_capi.OK, _capi.ITERATION_LIMIT)
        return rc, (x if has_x else None), float(fun.value), int(it.value), rows

    def k_generic_solve_f64(self, A, b, c, c0: float = 0.0, opts: "_capi.Opts | None" = None, want_log: bool = False):
        """Test hook: the generic (scalar-type-templated) kernels of the f32 instantiation, instantiated for double."""
        A, b, c = _f64(A), _f64(b), _f64(c)
        opts = opts or InteriorPoint.default().opts()
        m, n = A.shape
        x = np.full(n, np.nan)
        fun, it = C.c_double(np.nan), C.c_uint64(0)
        nlog = int(opts.max_iter) if want_log else 0
        log = (_capi.IterRow * max(nlog, 1))() if want_log else None
        rc = _capi.lib().lpipm_k_generic_solve_f64(self._h, m, n, _p(A), n, _p(b), _p(c), C.c_double(c0), C.byref(opts), _p(x),
                                                   C.byref(fun), C.byref(it), C.cast(log, C.c_void_p) if want_log else None)
        rows = [(r.alpha, r.rho_p, r.rho_d, r.rho_A, r.rho_g, r.rho_mu, r.obj) for r in list(log)[:min(int(it.value), nlog)]] if want_log else []
        return rc, x, fun.value, int(it.value), rows

    def set_profiling(self, on):
        """False/0 off, True/1 every phase, 2 only the A.D.A^T launches (2 events per iteration)."""
        _capi.lib().lpipm_set_profiling(self._h, int(on))

    def phase_times(self) -> dict:
        t = _capi.PhaseTimes()
        _capi.lib().lpipm_get_phase_times(self._h, C.byref(t))
        return {k: getattr(t, k) for k, _ in _capi.PhaseTimes._fields_}

    # ---- kernel-granularity entry points (parity tests / micro-benchmarks)
    def k_adat(self, dinv, repeats=1):
        dinv = _f64(dinv)
        M = np.empty((self.m, self.m))
        ms = C.c_double(0)
        _raise_for(_capi.lib().lpipm_k_adat(self._h, _p(dinv), _p(M), repeats, C.byref(ms)))
        return M, ms.value

    def k_potrf(self, M, repeats=1):
        L = _f64(M).copy()
        info, ms = C.c_int32(0), C.c_double(0)
        _raise_for(_capi.lib().lpipm_k_potrf(self._h, L.shape[0], _p(L), C.byref(info), repeats, C.byref(ms)))
        return L, info.value, ms.value

    def k_chol_solve(self, m, R, repeats=1):
        R = np.atleast_2d(_f64(R))
        V = np.empty_like(R)
        ms = C.c_double(0)
        _raise_for(_capi.lib().lpipm_k_chol_solve(self._h, m, R.shape[0], _p(R), _p(V), repeats, C.byref(ms)))
        return V, ms.value

    def k_symv_residual(self, M, V, R0):
        M = _f64(M); V = np.atleast_2d(_f64(V)); R0 = np.atleast_2d(_f64(R0))
        Rho = np.empty_like(V)
        _raise_for(_capi.lib().lpipm_k_symv_residual(self._h, M.shape[0], _p(M), V.shape[0], _p(V), _p(R0), _p(Rho)))
        return Rho

    def k_qr_solve(self, M, R):
        M = _f64(M)
        R = np.atleast_2d(_f64(R))
        V = np.empty_like(R)
        info, ms = C.c_int32(0), C.c_double(0)
        _raise_for(_capi.lib().lpipm_k_qr_solve(self._h, M.shape[0], _p(M), R.shape[0], _p(R), _p(V), C.byref(info),
                                                C.byref(ms)))
        return V, info.value, ms.value

    def k_gemv_n(self, W, repeats=1):
        W = np.atleast_2d(_f64(W))
        Y = np.empty((W.shape[0], self.m))
        ms = C.c_double(0)
        _raise_for(_capi.lib().lpipm_k_gemv_n(self._h, W.shape[0], _p(W), _p(Y), repeats, C.byref(ms)))
        return Y, ms.value

    def k_gemv_t(self, V, repeats=1):
        V = np.atleast_2d(_f64(V))
        U = np.empty((V.shape[0], self.n))
        ms = C.c_double(0)
        _raise_for(_capi.lib().lpipm_k_gemv_t(self._h, V.shape[0], _p(V), _p(U), repeats, C.byref(ms)))
        return U, ms.value

    def k_iteration(self, opts, x, y, z, tau, kappa, ip=False):
        """One loop body of solve_normal_form from the given iterate (lpipm_k_iteration).
        -> dict(x, y, z, tau, kappa, d_x, d_y, d_z, d_tau, d_kappa, alpha, info)"""
        x, y, z = _f64(x).copy(), _f64(y).copy(), _f64(z).copy()
        tk = np.array([float(tau), float(kappa)])
        dx, dy, dz, dtk, al = np.empty(self.n), np.empty(self.m), np.empty(self.n), np.empty(2), np.empty(1)
        info = C.c_int32(0)
        _raise_for(_capi.lib().lpipm_k_iteration(self._h, C.byref(opts), int(bool(ip)), _p(x), _p(y), _p(z), _p(tk[0:1]),
                                                 _p(tk[1:2]), _p(dx), _p(dy), _p(dz), _p(dtk), _p(al), C.byref(info)))
        return dict(x=x, y=y, z=z, tau=float(tk[0]), kappa=float(tk[1]), d_x=dx, d_y=dy, d_z=dz, d_tau=float(dtk[0]),
                    d_kappa=float(dtk[1]), alpha=float(al[0]), info=info.value)

    def k_gemv_dual(self, w, v, repeats=1):
        w, v = _f64(w), _f64(v)
        Aw, ATv = np.empty(self.m), np.empty(self.n)
        ms = C.c_double(0)
        _raise_for(_capi.lib().lpipm_k_gemv_dual(self._h, _p(w), _p(v), _p(Aw), _p(ATv), repeats, C.byref(ms)))
        return Aw, ATv, ms.value

    def k_mfma_f64_probe(self, iters=20000):
        tf, ms = C.c_double(0), C.c_double(0)
        _raise_for(_capi.lib().lpipm_k_mfma_f64_probe(self._h, iters, C.byref(tf), C.byref(ms)))
        return tf.value, ms.value


_default_ctx: dict[int, Context] = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


# ------------------------------------------------------------------------------ interior_point/mod.rs
class InteriorPointBuilder:
    """interior_point/mod.rs:41-138."""

    def __init__(self):
        o = _capi.Opts()
        _capi.lib().lpipm_default_opts(C.byref(o))   # mod.rs:50-60
        self._tol, self._disp, self._ip = o.tol, bool(o.disp), bool(o.ip)
        self._solver_type, self._alpha0, self._max_iter = EquationSolverType(o.solver_type), o.alpha0, o.max_iter

    def tol(self, tol):
        self._tol = float(tol)
        return self

    def disp(self, disp):
        self._disp = bool(disp)
        return self

    def ip(self, ip):
        self._ip = bool(ip)
        return self

    def solver_type(self, solver_type):
        self._solver_type = EquationSolverType(solver_type)
        return self

    def alpha0(self, alpha0):
        self._alpha0 = float(alpha0)
        return self

    def max_iter(self, max_iter):
        self._max_iter = int(max_iter)
        return self

    def build(self) -> "InteriorPoint":           # mod.rs:118-137
        if self._alpha0 <= 0.0 or self._alpha0 >= 1.0:
            raise InvalidParameter("Alpha0 must be between 0 and 1 (exclusive)")
        if self._tol <= 0.0:
            raise InvalidParameter("The tolerance must be nonnegative.")
        return InteriorPoint(self._tol, self._disp, self._ip, self._solver_type, self._alpha0, self._max_iter)


class InteriorPoint(Solver):
    """interior_point/mod.rs:140-241.  `device` selects the GPU (no reference analogue; default 0)."""

    def __init__(self, tol, disp, ip, solver_type, alpha0, max_iter, device: int = 0):
        self._tol, self._disp, self._ip = tol, disp, ip
        self._solver_type, self._alpha0, self._max_iter = solver_type, alpha0, max_iter
        self.device = device

    @staticmethod
    def default() -> "InteriorPoint":             # mod.rs:154-159
        return InteriorPointBuilder().build()

    @staticmethod
    def custom() -> InteriorPointBuilder:         # mod.rs:195-197
        return InteriorPointBuilder()

    def __eq__(self, other):                      # derive(PartialEq), mod.rs:140
        return isinstance(other, InteriorPoint) and self._key() == other._key()

    def _key(self):
        return (self._tol, self._disp, self._ip, self._solver_type, self._alpha0, self._max_iter)

    def opts(self) -> "_capi.Opts":
        return _capi.Opts(self._tol, self._alpha0, self._max_iter, int(self._ip), int(self._solver_type),
                          int(self._disp))

    def solve(self, problem: Problem) -> OptimizeResult:     # mod.rs:161-168
        ctx = default_context(self.device)
        if problem.dtype == np.float32:                      # InteriorPoint<f32> (src/float.rs:42-43)
            rc, x_slack, fun, it, _ = ctx.solve_f32(problem.A(), problem.b(), problem.c(), problem.c0(), self.opts())
            if rc == _capi.ITERATION_LIMIT:
                raise IterationLimitExceeded(x_slack)
            _raise_for(rc)
            return OptimizeResult(problem.denormalize_x_into(x_slack), fun, it)
        ctx.upload(problem)
        return self.solve_uploaded(ctx, problem)

    def solve_uploaded(self, ctx: Context, problem: Problem, want_log=False):
        rc, x_slack, fun, it, rows = ctx.solve_raw(self.opts(), want_log=want_log)
        if rc == _capi.ITERATION_LIMIT:
            raise IterationLimitExceeded(x_slack)            # mod.rs:237-239 (payload: x / tau, slack form)
        _raise_for(rc)
        res = OptimizeResult(problem.denormalize_x_into(x_slack), fun, it)   # mod.rs:165-167
        if want_log:
            res.log = rows
            res.x_slack = x_slack
        return res
