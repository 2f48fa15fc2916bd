// test_ripped.cpp -- the reference's own known-answer tests (src/lib.rs:77-114,
// src/solvers/interior_point/mod.rs:243-345) restated against include/ripped.hpp.
//   ./test_ripped host   : host-side logic only (no GPU): builders, validation, error mapping
//   ./test_ripped gpu    : everything, through liblpipm.so on device 0
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ripped.hpp"

using namespace ripped;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static bool close(const std::vector<double>& a, const std::vector<double>& b, double eps) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (!(std::fabs(a[i] - b[i]) <= eps)) return false;
    return true;
}

static Matrix mat(std::vector<double> d, uint64_t r, uint64_t c) { return Matrix{std::move(d), r, c}; }

static Problem make_problem() {  // src/lib.rs:84-96
    static const Matrix A_ub = mat({-3, 1, 1, 2}, 2, 2), A_eq = mat({1, 1}, 1, 2);
    static const std::vector<double> b_ub{6, 4}, b_eq{1}, c{-1, 4};
    return Problem::target(c).ub(A_ub, b_ub).eq(A_eq, b_eq).build();
}

static void host_tests() {
    // test_problem_interface, src/lib.rs:97-104 (+ slack form of linear_program.rs:145-160)
    Problem p = make_problem();
    CHECK(p.A().rows == 3 && p.A().cols == 4 && p.n_slack() == 2 && p.c0() == 0.0);
    CHECK(close(p.A().data, {-3, 1, 1, 0, 1, 2, 0, 1, 1, 1, 0, 0}, 0.0));
    CHECK(close(p.b(), {6, 4, 1}, 0.0) && close(p.c(), {-1, 4, 0, 0}, 0.0));
    // default_builder_doesnt_panic, mod.rs:249-254
    CHECK(InteriorPoint::default_() == InteriorPoint::custom().build());
    // InteriorPointBuilder::build validation, mod.rs:118-128
    for (double bad : {0.0, 1.0, -0.5, 2.0}) {
        try { InteriorPoint::custom().alpha0(bad).build(); CHECK(false); }
        catch (const LinearProgramError& e) { CHECK(e.kind() == ErrorKind::InvalidParameter); }
    }
    try { InteriorPoint::custom().tol(0.0).build(); CHECK(false); }
    catch (const LinearProgramError& e) { CHECK(e.kind() == ErrorKind::InvalidParameter); }
    // linear_program.rs:134-143
    std::vector<double> c{1, 2};
    try { Problem::target(c).build(); CHECK(false); }
    catch (const LinearProgramError& e) { CHECK(e.kind() == ErrorKind::Unconstrained); }
    Matrix bad = mat({1}, 1, 1);
    std::vector<double> bb{1};
    try { Problem::target(c).ub(bad, bb).build(); CHECK(false); }
    catch (const LinearProgramError& e) { CHECK(e.kind() == ErrorKind::IncompatibleInputDimensions); }
}

static void gpu_tests() {
    // test_interior_point_interface (src/lib.rs:106-113), test_interior_point_builder (mod.rs:256-273)
    {
        OptimizeResult res = InteriorPoint::default_().solve(make_problem());
        CHECK(close(res.x(), {1, 0}, 1e-6));
        CHECK(res.iteration() == 4 && std::fabs(res.fun() + 1.0) < 1e-6);
    }
    {   // crate doctest, src/lib.rs:39-51
        InteriorPoint s = InteriorPoint::custom().solver_type(EquationSolverType::Cholesky).tol(1e-8).disp(false)
                              .ip(true).alpha0(0.99995).max_iter(1000).build();
        CHECK(close(s.solve(make_problem()).x(), {1, 0}, 1e-6));
    }
    {   // InteriorPoint::custom doctest, mod.rs:175-194
        Matrix A_ub = mat({-3, 1, 1, 2}, 2, 2);
        std::vector<double> b_ub{6, 4}, c{-1, 4};
        CHECK(close(InteriorPoint::custom().build().solve(Problem::target(c).ub(A_ub, b_ub).build()).x(), {4, 0}, 1e-6));
    }
    Matrix A3 = mat({2, 1, 0, 0, 2, 1, 1, 0, 2}, 3, 3);
    std::vector<double> b3{1, 2, 3}, c3{-1, 4, -1.2};
    {   // test_linprog_eq_only, mod.rs:319-331
        CHECK(close(InteriorPoint::default_().solve(Problem::target(c3).eq(A3, b3).build()).x(),
                    {1.0 / 3, 1.0 / 3, 4.0 / 3}, 1e-6));
    }
    {   // test_linprog_ub_only, mod.rs:332-344
        CHECK(close(InteriorPoint::default_().solve(Problem::target(c3).ub(A3, b3).build()).x(), {0.5, 0.0, 1.25}, 1e-6));
    }
    {   // test_interior_point_inverse_solver / _least_squares_solver, mod.rs:275-317
        CHECK(close(InteriorPoint::custom().solver_type(EquationSolverType::Inverse).build().solve(make_problem()).x(),
                    {1, 0}, 1e-6));
        CHECK(close(InteriorPoint::custom().solver_type(EquationSolverType::LeastSquares).build().solve(make_problem()).x(),
                    {1, 0}, 1e-6));
    }
    {   // exits the reference defines but never tests: indicators.rs:66-83, mod.rs:232-239
        Matrix Ae = mat({1, 1}, 1, 2);
        std::vector<double> be{-1}, ce{1, 1};
        try { InteriorPoint::default_().solve(Problem::target(ce).eq(Ae, be).build()); CHECK(false); }
        catch (const LinearProgramError& e) { CHECK(e.kind() == ErrorKind::Infeasible); }
        Matrix Au = mat({1, -1}, 1, 2);
        std::vector<double> bu{0}, cu{-1, 0};
        try { InteriorPoint::default_().solve(Problem::target(cu).eq(Au, bu).build()); CHECK(false); }
        catch (const LinearProgramError& e) { CHECK(e.kind() == ErrorKind::Unbounded); }
        try { InteriorPoint::custom().max_iter(2).build().solve(make_problem()); CHECK(false); }
        catch (const LinearProgramError& e) { CHECK(e.kind() == ErrorKind::IterationLimitExceeded && e.best_x().size() == 4); }
    }
    {   // examples/symmetric.rs:10-25
        const uint64_t N = 1000;
        Matrix A; A.rows = A.cols = N; A.data.assign(N * N, 1.0);
        for (uint64_t i = 0; i < N; ++i) A.data[i * N + i] = 0.0;
        std::vector<double> b(N, (double)(N - 1)), c(N, -1.0);
        OptimizeResult r = InteriorPoint::custom().build().solve(Problem::target(c).ub(A, b).build());
        CHECK(close(r.x(), std::vector<double>(N, 1.0), 1e-10) && r.iteration() == 4);
    }
}

// The reference is generic over F: Float (src/float.rs:42-43); it holds no f32 test, so these restate lib.rs:84-113 in f32
// at a tolerance f32 can reach ("parity unpinned" by the reference; the f32 oracle is the checker in tests/test_gpu_f32.py).
static ProblemF32 make_problem_f32() {
    static const MatrixF32 A_ub{{-3, 1, 1, 2}, 2, 2}, A_eq{{1, 1}, 1, 2};
    static const std::vector<float> b_ub{6, 4}, b_eq{1}, c{-1, 4};
    return ProblemF32::target(c).ub(A_ub, b_ub).eq(A_eq, b_eq).build();
}
static void host_tests_f32() {
    ProblemF32 p = make_problem_f32();
    CHECK(p.A().rows == 3 && p.A().cols == 4 && p.n_slack() == 2 && p.c0() == 0.0f);
    const std::vector<float> want{-3, 1, 1, 0, 1, 2, 0, 1, 1, 1, 0, 0};
    CHECK(p.A().data == want);
    CHECK((p.b() == std::vector<float>{6, 4, 1}) && (p.c() == std::vector<float>{-1, 4, 0, 0}));
    std::vector<float> c{1, 2};
    try { ProblemF32::target(c).build(); CHECK(false); }
    catch (const LinearProgramError& e) { CHECK(e.kind() == ErrorKind::Unconstrained); }
}
static void gpu_tests_f32() {
    OptimizeResultF32 res = InteriorPoint::custom().tol(1e-4).build().solve(make_problem_f32());
    CHECK(res.x().size() == 2 && std::fabs(res.x()[0] - 1.0f) < 1e-3f && std::fabs(res.x()[1]) < 1e-3f);
    CHECK(std::fabs(res.fun() + 1.0f) < 1e-3f && res.iteration() >= 3 && res.iteration() <= 20);
    // at the default tolerance the f32 instantiation cannot pass the optimality test (6e-8 is f32's epsilon)
    try { InteriorPoint::custom().max_iter(40).build().solve(make_problem_f32()); CHECK(false); }
    catch (const LinearProgramError& e) {
        CHECK(e.kind() == ErrorKind::NumericalProblem || e.kind() == ErrorKind::IterationLimitExceeded);
    }
}

int main(int argc, char** argv) {
    host_tests();
    host_tests_f32();
    if (argc > 1 && !std::strcmp(argv[1], "gpu")) { gpu_tests(); gpu_tests_f32(); }
    std::printf(failures ? "%d FAILED\n" : "all ok\n", failures);
    return failures ? 1 : 0;
}
