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
// Matrices are row-major std::vector<F> + (rows, cols), the layout ndarray gives `Problem`.
// The reference is generic over `F: Float` (src/float.rs:8-43, f64 and f32): the classes below are templates over the
// scalar type with the f64 instantiation under the reference's plain names (Problem, Matrix, OptimizeResult) and the f32 one
// as ProblemF32 / MatrixF32 / OptimizeResultF32; InteriorPoint::solve takes either (f64: the hand-written fp64 path,
// f32: lpipm_solve_f32).
#pragma once
#include <type_traits>
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

template <class F> struct BasicMatrix {  // row-major, the layout of ndarray's standard Array2
    static_assert(std::is_same<F, double>::value || std::is_same<F, float>::value, "src/float.rs:33-43: f64 and f32");
    std::vector<F> data;
    uint64_t rows = 0, cols = 0;
};

template <class F> class BasicProblemBuilder;

// src/linear_program.rs:24-70
template <class F> class BasicProblem {
public:
    static BasicProblemBuilder<F> target(const std::vector<F>& c);   // :37-39
    const BasicMatrix<F>& A() const { return A_; }                   // :42-44
    const std::vector<F>& b() const { return b_; }                   // :47-49
    const std::vector<F>& c() const { return c_; }                   // :52-54
    F c0() const { return c0_; }                                     // :57-59
    uint64_t n_slack() const { return n_slack_; }
    std::vector<F> denormalize_x_into(std::vector<F> x_slack) const {  // :65-69
        x_slack.resize(x_slack.size() - n_slack_);
        return x_slack;
    }

private:
    friend class BasicProblemBuilder<F>;
    BasicMatrix<F> A_;
    std::vector<F> b_, c_;
    F c0_ = 0;
    uint64_t n_slack_ = 0;
};

// src/linear_program.rs:72-170
template <class F> class BasicProblemBuilder {
public:
    explicit BasicProblemBuilder(const std::vector<F>& c) : c_(c) {}
    BasicProblemBuilder& ub(const BasicMatrix<F>& A, const std::vector<F>& b) { ub_A_ = &A; ub_b_ = &b; return *this; }  // :93-96
    BasicProblemBuilder& eq(const BasicMatrix<F>& A, const std::vector<F>& b) { eq_A_ = &A; eq_b_ = &b; return *this; }  // :102-105
    BasicProblem<F> build() const {                                                                               // :125-169
        const uint64_t n = c_.size();
        const uint64_t m_ub = ub_A_ ? ub_A_->rows : 0, m_eq = eq_A_ ? eq_A_->rows : 0;
        if (m_ub + m_eq == 0) raise_for(LPIPM_UNCONSTRAINED);                                                     // :134-136
        if ((ub_A_ && (ub_A_->cols != n || ub_b_->size() != m_ub || ub_A_->data.size() != m_ub * n)) ||
            (eq_A_ && (eq_A_->cols != n || eq_b_->size() != m_eq || eq_A_->data.size() != m_eq * n)))
            raise_for(LPIPM_INCOMPATIBLE_DIMENSIONS);                                                             // :137-143
        BasicProblem<F> p;
        p.A_.rows = m_ub + m_eq;
        p.A_.cols = n + m_ub;
        p.A_.data.assign(p.A_.rows * p.A_.cols, F(0));
        p.b_.resize(p.A_.rows);
        p.c_.assign(p.A_.cols, F(0));
        if constexpr (std::is_same<F, double>::value) {
            raise_for(lpipm_problem_build(n, m_ub, ub_A_ ? ub_A_->data.data() : nullptr, ub_b_ ? ub_b_->data() : nullptr,
                                          m_eq, eq_A_ ? eq_A_->data.data() : nullptr, eq_b_ ? eq_b_->data() : nullptr,
                                          c_.data(), p.A_.data.data(), p.b_.data(), p.c_.data(), &p.n_slack_));
        } else {
            // the slack form of linear_program.rs:145-160 (the C ABI's lpipm_problem_build is the f64 instantiation):
            //   A = [[A_ub, I], [A_eq, 0]], b = [b_ub; b_eq], c = [c; 0]
            const uint64_t nn = p.A_.cols;
            for (uint64_t i = 0; i < m_ub; ++i) {
                for (uint64_t j = 0; j < n; ++j) p.A_.data[i * nn + j] = ub_A_->data[i * n + j];
                p.A_.data[i * nn + n + i] = F(1);
                p.b_[i] = (*ub_b_)[i];
            }
            for (uint64_t i = 0; i < m_eq; ++i) {
                for (uint64_t j = 0; j < n; ++j) p.A_.data[(m_ub + i) * nn + j] = eq_A_->data[i * n + j];
                p.b_[m_ub + i] = (*eq_b_)[i];
            }
            for (uint64_t j = 0; j < n; ++j) p.c_[j] = c_[j];
            p.n_slack_ = m_ub;
        }
        return p;
    }

private:
    const std::vector<F>& c_;
    const BasicMatrix<F>* ub_A_ = nullptr;
    const std::vector<F>* ub_b_ = nullptr;
    const BasicMatrix<F>* eq_A_ = nullptr;
    const std::vector<F>* eq_b_ = nullptr;
};
template <class F> inline BasicProblemBuilder<F> BasicProblem<F>::target(const std::vector<F>& c) { return BasicProblemBuilder<F>(c); }

// src/solvers/mod.rs:19-49
template <class F> class BasicOptimizeResult {
public:
    BasicOptimizeResult(std::vector<F> x, F fun, uint64_t iteration) : x_(std::move(x)), fun_(fun), iteration_(iteration) {}
    uint64_t iteration() const { return iteration_; }
    F fun() const { return fun_; }
    const std::vector<F>& x() const { return x_; }

private:
    std::vector<F> x_;
    F fun_;
    uint64_t iteration_;
};

using Matrix = BasicMatrix<double>;                 // the reference's Problem<f64>, ...
using Problem = BasicProblem<double>;
using ProblemBuilder = BasicProblemBuilder<double>;
using OptimizeResult = BasicOptimizeResult<double>;
using MatrixF32 = BasicMatrix<float>;               // ... and Problem<f32>
using ProblemF32 = BasicProblem<float>;
using ProblemBuilderF32 = BasicProblemBuilder<float>;
using OptimizeResultF32 = BasicOptimizeResult<float>;

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
    // The f32 instantiation of the same `solve` (src/float.rs:42-43): every operation in f32, lpipm_solve_f32.  With the
    // default tol = 1e-8 an f32 solve cannot satisfy the optimality test (the reference's behaviour): pass 1e-4 .. 1e-5.
    OptimizeResultF32 solve(const ProblemF32& problem) const {
        lpipm_ctx* ctx = nullptr;
        raise_for(lpipm_create(device_, &ctx));
        struct Guard { lpipm_ctx* c; ~Guard() { lpipm_destroy(c); } } guard{ctx};
        const MatrixF32& A = problem.A();
        std::vector<float> x(A.cols);
        float fun = 0.0f;
        uint64_t it = 0;
        const int rc = lpipm_solve_f32(ctx, A.rows, A.cols, A.data.data(), A.cols, problem.b().data(), problem.c().data(),
                                       problem.c0(), &o_, x.data(), &fun, &it, nullptr);
        if (rc == LPIPM_ITERATION_LIMIT) raise_for(rc, std::vector<double>(x.begin(), x.end()));
        raise_for(rc);
        return OptimizeResultF32(problem.denormalize_x_into(std::move(x)), fun, it);
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
