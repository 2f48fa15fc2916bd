// ripped.hpp -- header-only C++17 host-side mirror of the public API of sebasv/lp (crate `ripped`
// 0.1.1) for the one path liblpipm.so accelerates, written over the C ABI of lpipm.h.
//
// The reference is compiled Rust and this image has no Rust toolchain, so the host side above the
// C ABI is mirrored in C++ (and in Python, lp_amd/__init__.py): same names, same argument meaning,
// same error behaviour.  A Rust `Result::Err(e)` is a thrown ripped::LinearProgramError whose
// `kind()` is the variant of src/error.rs:10-28.
//
//   reference (Rust)                                      here
//   Problem::target(&c).ub(&A,&b).eq(&A,&b).build()?      Problem::target(c).ub(A,b).eq(A,b).build()
//   InteriorPoint::custom().tol(1e-8)....build()?          InteriorPoint::custom().tol(1e-8)....build()
//   solver.solve(&problem)? -> OptimizeResult              solver.solve(problem) -> OptimizeResult
//   res.x() / res.fun() / res.iteration()                  res.x() / res.fun() / res.iteration()
//
// Matrices are row-major std::vector<double> + (rows, cols), the layout ndarray gives `Problem`.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "lpipm.h"

namespace ripped {

// src/error.rs:10-28
enum class ErrorKind {
    Unconstrained = LPIPM_UNCONSTRAINED,
    NumericalProblem = LPIPM_NUMERICAL_PROBLEM,
    InvalidParameter = LPIPM_INVALID_PARAMETER,
    IncompatibleInputDimensions = LPIPM_INCOMPATIBLE_DIMENSIONS,
    Infeasible = LPIPM_INFEASIBLE,
    Unbounded = LPIPM_UNBOUNDED,
    IterationLimitExceeded = LPIPM_ITERATION_LIMIT,
    Backend = LPIPM_ERR_HIP  // HIP / driver failure: no analogue in the reference, never swallowed
};

class LinearProgramError : public std::runtime_error {
public:
    LinearProgramError(ErrorKind k, const std::string& msg, std::vector<double> x = {})
        : std::runtime_error(msg), kind_(k), x_(std::move(x)) {}
    ErrorKind kind() const { return kind_; }
    // IterationLimitExceeded(x / tau): error.rs:26-28, interior_point/mod.rs:237-239
    const std::vector<double>& best_x() const { return x_; }

private:
    ErrorKind kind_;
    std::vector<double> x_;
};

inline void raise_for(int status, std::vector<double> x = {}) {
    if (status == LPIPM_OK) return;
    std::string msg = lpipm_strerror(status);
    if (status >= 100) {
        msg += ": ";
        msg += lpipm_last_error_detail();
        throw LinearProgramError(ErrorKind::Backend, msg);
    }
    throw LinearProgramError(static_cast<ErrorKind>(status), msg, std::move(x));
}

struct Matrix {  // row-major, the layout of ndarray's standard Array2
    std::vector<double> data;
    uint64_t rows = 0, cols = 0;
};

class ProblemBuilder;

// src/linear_program.rs:24-70
class Problem {
public:
    static ProblemBuilder target(const std::vector<double>& c);  // :37-39
    const Matrix& A() const { return A_; }                       // :42-44
    const std::vector<double>& b() const { return b_; }          // :47-49
    const std::vector<double>& c() const { return c_; }          // :52-54
    double c0() const { return c0_; }                            // :57-59
    uint64_t n_slack() const { return n_slack_; }
    std::vector<double> denormalize_x_into(std::vector<double> x_slack) const {  // :65-69
        x_slack.resize(x_slack.size() - n_slack_);
        return x_slack;
    }

private:
    friend class ProblemBuilder;
    Matrix A_;
    std::vector<double> b_, c_;
    double c0_ = 0.0;
    uint64_t n_slack_ = 0;
};

// src/linear_program.rs:72-170
class ProblemBuilder {
public:
    explicit ProblemBuilder(const std::vector<double>& c) : c_(c) {}
    ProblemBuilder& ub(const Matrix& A, const std::vector<double>& b) { ub_A_ = &A; ub_b_ = &b; return *this; }  // :93-96
    ProblemBuilder& eq(const Matrix& A, const std::vector<double>& b) { eq_A_ = &A; eq_b_ = &b; return *this; }  // :102-105
    Problem build() const {                                                                                       // :125-169
        const uint64_t n = c_.size();
        const uint64_t m_ub = ub_A_ ? ub_A_->rows : 0, m_eq = eq_A_ ? eq_A_->rows : 0;
        if (m_ub + m_eq == 0) raise_for(LPIPM_UNCONSTRAINED);                                                     // :134-136
        if ((ub_A_ && (ub_A_->cols != n || ub_b_->size() != m_ub || ub_A_->data.size() != m_ub * n)) ||
            (eq_A_ && (eq_A_->cols != n || eq_b_->size() != m_eq || eq_A_->data.size() != m_eq * n)))
            raise_for(LPIPM_INCOMPATIBLE_DIMENSIONS);                                                             // :137-143
        Problem p;
        p.A_.rows = m_ub + m_eq;
        p.A_.cols = n + m_ub;
        p.A_.data.resize(p.A_.rows * p.A_.cols);
        p.b_.resize(p.A_.rows);
        p.c_.resize(p.A_.cols);
        raise_for(lpipm_problem_build(n, m_ub, ub_A_ ? ub_A_->data.data() : nullptr, ub_b_ ? ub_b_->data() : nullptr,
                                      m_eq, eq_A_ ? eq_A_->data.data() : nullptr, eq_b_ ? eq_b_->data() : nullptr,
                                      c_.data(), p.A_.data.data(), p.b_.data(), p.c_.data(), &p.n_slack_));
        return p;
    }

private:
    const std::vector<double>& c_;
    const Matrix* ub_A_ = nullptr;
    const std::vector<double>* ub_b_ = nullptr;
    const Matrix* eq_A_ = nullptr;
    const std::vector<double>* eq_b_ = nullptr;
};
inline ProblemBuilder Problem::target(const std::vector<double>& c) { return ProblemBuilder(c); }

// src/solvers/mod.rs:19-49
class OptimizeResult {
public:
    OptimizeResult(std::vector<double> x, double fun, uint64_t iteration)
        : x_(std::move(x)), fun_(fun), iteration_(iteration) {}
    uint64_t iteration() const { return iteration_; }
    double fun() const { return fun_; }
    const std::vector<double>& x() const { return x_; }

private:
    std::vector<double> x_;
    double fun_;
    uint64_t iteration_;
};

enum class EquationSolverType { Cholesky = 0, Inverse = 1, LeastSquares = 2 };  // newton_equations.rs:37-46

class InteriorPoint;

// src/solvers/interior_point/mod.rs:41-138
class InteriorPointBuilder {
public:
    InteriorPointBuilder() { lpipm_default_opts(&o_); }  // mod.rs:50-60
    InteriorPointBuilder& tol(double v) { o_.tol = v; return *this; }
    InteriorPointBuilder& disp(bool v) { o_.disp = v; return *this; }
    InteriorPointBuilder& ip(bool v) { o_.ip = v; return *this; }
    InteriorPointBuilder& solver_type(EquationSolverType v) { o_.solver_type = static_cast<int32_t>(v); return *this; }
    InteriorPointBuilder& alpha0(double v) { o_.alpha0 = v; return *this; }
    InteriorPointBuilder& max_iter(uint64_t v) { o_.max_iter = v; return *this; }
    InteriorPoint build() const;  // mod.rs:118-137

private:
    lpipm_opts o_;
};

// src/solvers/mod.rs:12-16 (trait Solver) + interior_point/mod.rs:140-241
class InteriorPoint {
public:
    explicit InteriorPoint(const lpipm_opts& o, int device = 0) : o_(o), device_(device) {}
    static InteriorPoint default_() { return InteriorPointBuilder().build(); }  // mod.rs:154-159
    static InteriorPointBuilder custom() { return InteriorPointBuilder(); }     // mod.rs:195-197
    bool operator==(const InteriorPoint& r) const {                             // derive(PartialEq), mod.rs:140
        return o_.tol == r.o_.tol && o_.alpha0 == r.o_.alpha0 && o_.max_iter == r.o_.max_iter && o_.ip == r.o_.ip &&
               o_.solver_type == r.o_.solver_type && o_.disp == r.o_.disp;
    }
    const lpipm_opts& opts() const { return o_; }

    // mod.rs:161-168.  One context (stream + device buffers) per call keeps `solve(&self)` stateless
    // and re-entrant like the reference; callers that solve many problems can hold an lpipm_ctx.
    OptimizeResult solve(const Problem& problem) const {
        lpipm_ctx* ctx = nullptr;
        raise_for(lpipm_create(device_, &ctx));
        struct Guard { lpipm_ctx* c; ~Guard() { lpipm_destroy(c); } } guard{ctx};
        const Matrix& A = problem.A();
        // n_slack tells the backend that the last columns are the [I; 0] slack block (linear_program.rs:147-161)
        raise_for(lpipm_upload_slack(ctx, A.rows, A.cols, A.data.data(), A.cols, problem.b().data(),
                                     problem.c().data(), problem.c0(), problem.n_slack()));
        std::vector<double> x(A.cols);
        double fun = 0.0;
        uint64_t it = 0;
        const int rc = lpipm_solve(ctx, &o_, x.data(), &fun, &it, nullptr);
        if (rc == LPIPM_ITERATION_LIMIT) raise_for(rc, x);          // payload: x / tau, mod.rs:237-239
        raise_for(rc);
        return OptimizeResult(problem.denormalize_x_into(std::move(x)), fun, it);  // mod.rs:165-167
    }

private:
    lpipm_opts o_;
    int device_;
};

inline InteriorPoint InteriorPointBuilder::build() const {
    if (!(o_.alpha0 > 0.0) || !(o_.alpha0 < 1.0))   // mod.rs:119-123
        throw LinearProgramError(ErrorKind::InvalidParameter,
                                 "A parameter was set to an invalid value: Alpha0 must be between 0 and 1 (exclusive)");
    if (!(o_.tol > 0.0))                            // mod.rs:124-128
        throw LinearProgramError(ErrorKind::InvalidParameter,
                                 "A parameter was set to an invalid value: The tolerance must be nonnegative.");
    return InteriorPoint(o_);
}

}  // namespace ripped
