/*
 * oracle_ipm.h -- CPU restatement of the sebasv/lp ("ripped" 0.1.1) interior-point hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline.  The product (lp_amd/, liblpipm.so) never links,
 * imports or calls it.
 *
 * Pinning: the reference is Rust and neither cargo nor rustc exists in the build container
 * (nothing was denied; the toolchain is absent), so there is no oracle/_ref build.  The
 * restatement is pinned END-TO-END by every known-answer test the reference holds for this path
 * (tests/test_oracle_golden.py: src/lib.rs:23-51,106-113; interior_point/mod.rs:175-194,256-344;
 * examples/symmetric.rs:10-25).  The reference has no fixture for M, the Cholesky factor or a
 * triangular solve (its arithmetic lives in the un-vendored crates ndarray 0.15 /
 * matrixmultiply / linfa-linalg 0.1, Cargo.toml:31-36, no Cargo.lock), so parity at kernel
 * granularity is "unpinned" and anchored on the call sites cited in oracle_ipm.c.
 */
#ifndef ORACLE_IPM_H
#define ORACLE_IPM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* interior_point/mod.rs:41-48 (InteriorPointBuilder fields) */
typedef struct {
    double   tol;         /* mod.rs:53  default 1e-8    */
    double   alpha0;      /* mod.rs:57  default 0.99995 */
    uint64_t max_iter;    /* mod.rs:58  default 1000    */
    int32_t  ip;          /* mod.rs:55  default true    */
    int32_t  solver_type; /* mod.rs:56  0 Cholesky, 1 Inverse, 2 LeastSquares */
    int32_t  disp;        /* mod.rs:54  default false   */
} oracle_opts;

/* indicators.rs:8-23 + the alpha printed at mod.rs:228 */
typedef struct { double alpha, rho_p, rho_d, rho_A, rho_g, rho_mu, obj; } oracle_iter_row;

/* per-phase wall time of one solve, seconds (CPU-baseline reporting only) */
typedef struct { double adat, chol, solves, gemv, rest, total; } oracle_timing;

/* error.rs:10-28 mapped to integers (0 = Ok) */
enum {
    ORACLE_OK = 0, ORACLE_UNCONSTRAINED = 1, ORACLE_NUMERICAL_PROBLEM = 2,
    ORACLE_INVALID_PARAMETER = 3, ORACLE_INCOMPATIBLE_DIMENSIONS = 4, ORACLE_INFEASIBLE = 5,
    ORACLE_UNBOUNDED = 6, ORACLE_ITERATION_LIMIT = 7
};

void oracle_default_opts(oracle_opts* o);

/* linear_program.rs:125-169: slack form [[A_ub I],[A_eq 0]], b=[b_ub;b_eq], c=[c;0].
 * Outputs must hold (m_ub+m_eq)*(n+m_ub), (m_ub+m_eq), (n+m_ub) doubles. */
int oracle_problem_build(uint64_t n, uint64_t m_ub, const double* A_ub, const double* b_ub,
                         uint64_t m_eq, const double* A_eq, const double* b_eq, const double* c,
                         double* A_out, double* b_out, double* c_out, uint64_t* n_slack_out);

/* interior_point/mod.rs:199-240 on the slack-form problem (A m x n row-major, lda = n).
 * x_slack_out[n] = x/tau (also filled for ORACLE_ITERATION_LIMIT, mod.rs:237-239);
 * fun_out = c.x + c0 (linear_program.rs:61-63); log (nullable) gets one row per iteration. */
int oracle_ipm_solve(uint64_t m, uint64_t n, const double* A, const double* b, const double* c,
                     double c0, const oracle_opts* opts, double* x_slack_out, double* fun_out,
                     uint64_t* iterations_out, oracle_iter_row* log, oracle_timing* timing);

/* one loop body of solve_normal_form from a given iterate (see oracle_ipm.c) */
int oracle_iteration(uint64_t m, uint64_t n, const double* A, const double* b, const double* c, int solver_type,
                     int ip, double alpha0, double* x, double* y, double* z, double* tau, double* kappa,
                     double* d_x, double* d_y, double* d_z, double* d_tk, double* alpha_out);

/* kernel-granularity restatements used by the per-kernel differential tests */
/* newton_equations.rs:54-57: M = A . (Dinv[:,None] * A^T), full square, n x m temporary */
void oracle_adat(uint64_t m, uint64_t n, const double* A, const double* dinv, double* M);
/* newton_equations.rs:129-131: lower Cholesky factor of M (row-major, in place, upper zeroed);
 * returns 0, or k+1 for the first non-positive pivot k */
int  oracle_cholesky(uint64_t m, double* M);
/* newton_equations.rs:151-169: v = L^-T (L^-1 r) */
void oracle_cholesky_solve(uint64_t m, const double* L, const double* r, double* v);
void oracle_gemv_n(uint64_t m, uint64_t n, const double* A, const double* w, double* y);
void oracle_gemv_t(uint64_t m, uint64_t n, const double* A, const double* v, double* u);

#ifdef __cplusplus
}
#endif
#endif
