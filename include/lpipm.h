/*
 * lpipm.h -- C ABI of the MI355X-native interior-point LP hot path (liblpipm.so).
 *
 * Drop-in boundary for the ONE hot path of sebasv/lp (crate `ripped` 0.1.1):
 * `InteriorPoint::solve()` on a slack-form `Problem` -- forming A.diag(x/z).A^T, its Cholesky
 * factor, the triangular solves and the residual / direction GEMVs of the homogeneous self-dual
 * Mehrotra predictor-corrector algorithm -- as hand-written HIP kernels for gfx950.
 *
 * The reference has no FFI; each entry point below names the reference interface it replaces
 * (paths relative to /root/reference/src).  Plain pointers and sizes only; no C++/torch types.
 * The reference-side binding a maintainer would add (a `hip` feature arm next to
 * `cfg(feature = "blas")`, newton_equations.rs:2-13) is shown in INTEGRATION.md and
 * bindings/rust/.
 *
 * Conventions
 *   - every function returns an lpipm_status (0 = Ok) unless stated otherwise;
 *   - the caller owns every host pointer; the library copies in, never retains a host pointer
 *     after return, never calls back, never unwinds across the boundary;
 *   - a ctx is bound to one device and one stream and is not thread-safe; distinct ctxs are
 *     independent (one per GPU for batches);
 *   - matrices are row-major fp64, exactly as `Problem` holds them (ndarray standard layout,
 *     linear_program.rs:145-156).
 */
#ifndef LPIPM_H
#define LPIPM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error.rs:10-28 (LinearProgramError) as integers; >= 100 have no reference analogue. */
typedef enum {
    LPIPM_OK                      = 0,
    LPIPM_UNCONSTRAINED           = 1, /* error.rs:12  */
    LPIPM_NUMERICAL_PROBLEM       = 2, /* error.rs:15  */
    LPIPM_INVALID_PARAMETER       = 3, /* error.rs:18  */
    LPIPM_INCOMPATIBLE_DIMENSIONS = 4, /* error.rs:21  */
    LPIPM_INFEASIBLE              = 5, /* error.rs:24  */
    LPIPM_UNBOUNDED               = 6, /* error.rs:27  */
    LPIPM_ITERATION_LIMIT         = 7, /* error.rs:28  IterationLimitExceeded(x / tau): x_out IS filled */
    LPIPM_ERR_HIP                 = 100, /* HIP runtime failure (lpipm_last_error_detail has the text) */
    LPIPM_ERR_NO_PROBLEM          = 101, /* solve before upload */
    LPIPM_ERR_UNSUPPORTED         = 102, /* valid in the reference, not built here (QR arms beyond m = 16384) */
    LPIPM_ERR_BAD_ARGUMENT        = 103  /* null pointer / lda < n / size overflow */
} lpipm_status;

/* solvers/interior_point/newton_equations.rs:37-46 (EquationSolverType) */
enum { LPIPM_SOLVER_CHOLESKY = 0, LPIPM_SOLVER_INVERSE = 1, LPIPM_SOLVER_LEAST_SQUARES = 2 };

/* solvers/interior_point/mod.rs:41-48 (InteriorPointBuilder / InteriorPoint fields) */
typedef struct {
    double   tol;         /* mod.rs:53  default 1e-8    ; must be > 0        (mod.rs:124) */
    double   alpha0;      /* mod.rs:57  default 0.99995 ; 0 < alpha0 < 1     (mod.rs:119) */
    uint64_t max_iter;    /* mod.rs:58  default 1000 */
    int32_t  ip;          /* mod.rs:55  default 1 (alternative initial point) */
    int32_t  solver_type; /* mod.rs:56  default LPIPM_SOLVER_CHOLESKY */
    int32_t  disp;        /* mod.rs:54  default 0; 1 prints the reference's table (mod.rs:208-211,227-229) */
} lpipm_opts;

/* One row of the `disp` table: alpha (mod.rs:228) + Indicators (indicators.rs:8-23). */
typedef struct { double alpha, rho_p, rho_d, rho_A, rho_g, rho_mu, obj; } lpipm_iter_row;
/* The same row of an f32 solve (lpipm_solve_f32). */
typedef struct { float alpha, rho_p, rho_d, rho_A, rho_g, rho_mu, obj; } lpipm_iter_row_f32;

/* Device time per phase of the LAST lpipm_solve on this ctx, from HIP events on the ctx's stream
 * (only filled while profiling is on; recording events costs a few us per phase). */
typedef struct {
    double   adat_ms;      /* sum over iterations of the A.D.A^T kernel launches       */
    double   potrf_ms;     /* Cholesky factorisation (+ diagonal-block inverses)        */
    double   trsv_ms;      /* triangular solves                                        */
    double   gemv_ms;      /* passes over A (GEMV-N / GEMV-T)                           */
    double   vec_ms;       /* O(n) vector / scalar kernels + status read-back          */
    double   total_ms;     /* first kernel to last kernel of the solve                 */
    uint64_t adat_launches;
    uint64_t iterations;
    uint64_t gemv_passes;  /* passes over A inside gemv_ms (a 2-vector pass reads A once and counts once) */
} lpipm_phase_times;

typedef struct lpipm_ctx lpipm_ctx; /* opaque: device buffers + stream for one (thread, device) */

/* InteriorPointBuilder::new (mod.rs:50-60) */
void lpipm_default_opts(lpipm_opts* out);
/* Display strings of error.rs:10-28 (+ the >= 100 codes).  Never NULL. */
const char* lpipm_strerror(int status);
/* Text of the last HIP failure on this thread ("" if none). */
const char* lpipm_last_error_detail(void);
/* Number of visible HIP devices (0 when there is no GPU or no driver). */
int lpipm_device_count(void);

/* ProblemBuilder::build (linear_program.rs:125-169): slack form
 *   A = [[A_ub I],[A_eq 0]]  ((m_ub+m_eq) x (n+m_ub)),  b = [b_ub; b_eq],  c = [c; 0].
 * Host-side, O(mn), once per problem.  A_ub / A_eq may be NULL when their row count is 0.
 * Outputs must hold (m_ub+m_eq)*(n+m_ub), (m_ub+m_eq) and (n+m_ub) doubles. */
int lpipm_problem_build(uint64_t n, uint64_t m_ub, const double* A_ub, const double* b_ub,
                        uint64_t m_eq, const double* A_eq, const double* b_eq, const double* c,
                        double* A_out, double* b_out, double* c_out, uint64_t* n_slack_out);

/* One context per (thread, device).  Fails with LPIPM_ERR_HIP when the device is unusable:
 * there is NO CPU fallback anywhere behind this ABI. */
int  lpipm_create(int device, lpipm_ctx** out);
void lpipm_destroy(lpipm_ctx* ctx);

/* Upload the slack-form problem `Problem` holds (accessors A() b() c(), linear_program.rs:42-59;
 * c0 :56-59).  A is m x n row-major with leading dimension lda >= n.  One H2D copy of A; the
 * hot loop never touches host memory again except a 96-byte status read-back per iteration. */
int lpipm_upload(lpipm_ctx* ctx, uint64_t m, uint64_t n, const double* A, uint64_t lda,
                 const double* b, const double* c, double c0);

/* Same upload with the structural hint `Problem` carries (n_slack, linear_program.rs:161): the last
 * n_slack columns of A are the slack block [I; 0] of `ub` constraints (linear_program.rs:147-156).
 * They are then neither copied to the device nor multiplied: M = A_x D_x A_x^T + diag(D_s), A.w adds
 * w_s, A^T.v copies v.  The hint is verified; a matrix without that structure is treated as dense.
 * Results agree with lpipm_upload to rounding (same iteration counts on every test). */
int lpipm_upload_slack(lpipm_ctx* ctx, uint64_t m, uint64_t n, const double* A, uint64_t lda,
                       const double* b, const double* c, double c0, uint64_t n_slack);

/* Device-side assembly of the slack form (ProblemBuilder::build, linear_program.rs:125-169, without its
 * (m_ub+m_eq) x (n+m_ub) host matrix): the `ub` and `eq` blocks go to the device as they are -- rows of A_ub
 * first, then rows of A_eq (linear_program.rs:145-156) -- b = [b_ub; b_eq] (:157-158), c = [c; 0] (:159-160),
 * and the slack block [I; 0] is never formed anywhere.  Equivalent to lpipm_problem_build + lpipm_upload_slack
 * (bit-identical solves); x_slack_out of lpipm_solve then has n + m_ub entries, slack values last.
 * Either block may be absent (m_ub or m_eq 0, pointer NULL); both absent -> LPIPM_UNCONSTRAINED (:134-136). */
int lpipm_upload_ub_eq(lpipm_ctx* ctx, uint64_t n, uint64_t m_ub, const double* A_ub, uint64_t lda_ub,
                       const double* b_ub, uint64_t m_eq, const double* A_eq, uint64_t lda_eq, const double* b_eq,
                       const double* c, double c0);

/* InteriorPoint::solve_normal_form + the `fun` of solve (mod.rs:199-240, :165).
 *   x_slack_out[n] : x / tau  (mod.rs:231); ALSO filled for LPIPM_ITERATION_LIMIT (mod.rs:237-239)
 *   fun_out        : c . x_slack + c0  (linear_program.rs:61-63)
 *   iterations_out : iteration at which the status was decided (mod.rs:213,231)
 *   log            : nullable; one row per iteration, at most max_iter rows
 * Dropping the n_slack tail (denormalize_x_into, linear_program.rs:65-69) stays with the caller,
 * where the reference has it (mod.rs:166).  Option validation mirrors mod.rs:118-128. */
int lpipm_solve(lpipm_ctx* ctx, const lpipm_opts* opts, double* x_slack_out, double* fun_out,
                uint64_t* iterations_out, lpipm_iter_row* log);

/* Same solve, solution left in HBM: copies x / tau (n doubles) to a DEVICE pointer on the ctx's
 * stream (for the one RCCL gather of a sharded batch); x_dev_out may be NULL. */
int lpipm_solve_device(lpipm_ctx* ctx, const lpipm_opts* opts, void* x_dev_out, double* fun_out,
                       uint64_t* iterations_out, lpipm_iter_row* log);

/* A shard of independent LPs on ONE device (BASELINE config 4): problem i is m[i] x n[i] with
 * A[i] (lda = n[i]), b[i], c[i], c0[i]; results go to x_slack_out[i] (n[i] doubles), fun_out[i],
 * iterations_out[i], status_out[i].  Returns the first non-Ok *runtime* status (>= 100) or Ok;
 * per-problem solver outcomes (Infeasible, ...) are reported in status_out only.
 * Members of equal shape are solved as lockstep batches (below), the others one at a time on worker contexts.
 * The call replaces the context's uploaded problem: upload again before a later lpipm_solve on this context. */
int lpipm_solve_batch(lpipm_ctx* ctx, uint64_t count, const uint64_t* m, const uint64_t* n,
                      const double* const* A, const double* const* b, const double* const* c,
                      const double* c0, const lpipm_opts* opts, double* const* x_slack_out,
                      double* fun_out, uint64_t* iterations_out, int32_t* status_out);

/* ---- one LP split by COLUMNS over several ranks / GPUs (BASELINE config C5) -----------------------------
 * Rank g holds the column block A[:, J_g] (m x n_local), c[J_g] and the matching slices of x, z; b and y
 * are replicated.  Per iteration the partial normal equations M_g = A_g D_g A_g^T are summed over ranks
 * (the "all-reduce on A.D.A^T panels"), as are A_g.w_g (m doubles) and a handful of dot products / ratio-test
 * minima; the factorisation runs replicated on every rank.  The library never links a communication
 * library: the caller supplies the all-reduce (RCCL `ncclAllReduce` on `stream` from Rust/C++, a
 * torch.distributed call from the Python harness).
 *   fn(user, dev_ptr, count, op, stream): in-place all-reduce of `count` doubles at device pointer dev_ptr,
 *   op 0 = sum, 1 = min; the stream has been drained before the call; return 0 when the result is in place. */
typedef int (*lpipm_allreduce_fn)(void* user, void* dev_ptr, uint64_t count, int op, void* stream);
int lpipm_set_collective(lpipm_ctx* ctx, int rank, int world, lpipm_allreduce_fn fn, void* user);
/* on = 1: the library no longer drains its stream before calling fn; fn must ENQUEUE the reduction on the stream it
 * is handed (RCCL: ncclAllReduce(ptr, ptr, count, ncclDouble, op, comm, (hipStream_t)stream)) and may return at once --
 * stream order makes the result visible to the kernels that follow, and nothing on the host waits (the one
 * status read-back per iteration excepted).  on = 0 (default): the drained contract described above. */
int lpipm_set_collective_on_stream(lpipm_ctx* ctx, int on);
/* Upload this rank's column block; afterwards lpipm_solve / lpipm_solve_device run the n-split algorithm and
 * return this rank's slice of x / tau (n_local doubles); fun, iterations and the log are global. */
int lpipm_upload_nsplit(lpipm_ctx* ctx, uint64_t m, uint64_t n_total, uint64_t n_local, const double* A_local,
                        uint64_t lda, const double* b, const double* c_local, double c0);

/* Lockstep batch: `count` LPs of ONE shape (m x n, dense) resident on the device at once; every kernel launch of
 * the iteration covers all of them, so the ~100 dependent launches per iteration are paid once per batch
 * instead of once per LP.  lpipm_upload_lockstep replaces the context's problem; lpipm_solve_lockstep returns
 * per-LP status (0 / LinearProgramError variants), fun, iterations and x / tau, exactly as `count` calls of
 * lpipm_solve would (Cholesky arm only: LPIPM_ERR_UNSUPPORTED otherwise).  A[i]: m x n row-major, lda = n. */
int lpipm_upload_lockstep(lpipm_ctx* ctx, uint64_t count, uint64_t m, uint64_t n, const double* const* A,
                          const double* const* b, const double* const* c, const double* c0 /* nullable */);
int lpipm_solve_lockstep(lpipm_ctx* ctx, const lpipm_opts* opts, double* const* x_slack_out, double* fun_out,
                         uint64_t* iterations_out, int32_t* status_out);
/* The same with the solutions left in HBM: x / tau of LP i goes to the DEVICE row x_dev_out + i * row_stride doubles
 * (row_stride >= n), device to device on the ctx's stream -- the packed block a sharded batch then all-gathers (RCCL)
 * without a host round trip.  Rows of members without a solution (Infeasible, ...) are left untouched. */
int lpipm_solve_lockstep_device(lpipm_ctx* ctx, const lpipm_opts* opts, void* x_dev_out, uint64_t row_stride,
                                double* fun_out, uint64_t* iterations_out, int32_t* status_out);
/* lpipm_solve_batch with device-resident results: member i's x / tau goes to x_dev_out + i * row_stride doubles
 * (row_stride >= max n[i]); everything else as lpipm_solve_batch. */
int lpipm_solve_batch_device(lpipm_ctx* ctx, uint64_t count, const uint64_t* m, const uint64_t* n,
                             const double* const* A, const double* const* b, const double* const* c,
                             const double* c0 /* nullable */, const lpipm_opts* opts, void* x_dev_out,
                             uint64_t row_stride, double* fun_out /* nullable */, uint64_t* iterations_out /* nullable */,
                             int32_t* status_out);
/* lpipm_solve_batch groups members of equal shape into lockstep batches: max_group -1 = auto (default: chunks
 * of up to 32 within the memory budget, the upload of one chunk overlapping the solve of the previous one), 0 = never, > 0 = largest group. */
int lpipm_set_batch_lockstep(lpipm_ctx* ctx, int max_group);

/* Number of LPs of a batch in flight at once on the device (0 = auto, the default: 8 for members up
 * to m = 2048, else 2; 1 = strictly one after the other).  Members of a batch are independent, each in-flight member has its own stream and buffers. */
int lpipm_set_batch_concurrency(lpipm_ctx* ctx, int nworkers);

/* Profiling switch (off by default) and the per-phase device times of the last solve: HIP events on the
 * context's own stream.  on = 1: every phase (~14 events per iteration); on = 2: only the A.D.A^T launches
 * (2 events per iteration; the other phases are lumped into vec_ms). */
int lpipm_set_profiling(lpipm_ctx* ctx, int on);
int lpipm_get_phase_times(const lpipm_ctx* ctx, lpipm_phase_times* out);

/* ---- kernel-granularity entry points ---------------------------------------------------------
 * Each runs ONE stage of the hot path on the uploaded problem / the given operands so that the
 * parity tests can compare it with the oracle's restatement of the cited reference lines.
 * Host pointers in, host pointers out (copies are outside any timing the library reports). */

/* newton_equations.rs:54-57:  M = A . diag(dinv) . A^T.  dinv[n]; M_out m x m row-major, LOWER
 * triangle valid (the strict upper triangle is unspecified).  `ms_out` (nullable) gets the device
 * time of the kernel launch(es) averaged over `repeats` (>= 1) back-to-back launches. */
int lpipm_k_adat(lpipm_ctx* ctx, const double* dinv, double* M_out, int repeats, double* ms_out);
/* newton_equations.rs:129-131: in-place lower Cholesky of the m x m row-major matrix M (lower
 * triangle read, lower triangle written).  info_out: 0, or k+1 for the first non-positive pivot. */
int lpipm_k_potrf(lpipm_ctx* ctx, uint64_t m, double* M_inout, int32_t* info_out, int repeats,
                  double* ms_out);
/* newton_equations.rs:151-169: V[r] = L^-T L^-1 R[r] for nrhs (1 or 2) right-hand sides, with
 * L = the factor of a preceding lpipm_k_potrf on this ctx (kept on device).  R, V: nrhs x m. */
int lpipm_k_chol_solve(lpipm_ctx* ctx, uint64_t m, int nrhs, const double* R, double* V,
                       int repeats, double* ms_out);
/* The residual of the refinement step of the Cholesky solve (solver.hip chol_solve_refined): Rho[q] = R0[q] - M.V[q],
 * q < nrhs (1|2), for a symmetric m x m row-major M of which only the LOWER triangle is read.  V, R0, Rho: nrhs x m. */
int lpipm_k_symv_residual(lpipm_ctx* ctx, uint64_t m, const double* M, int nrhs, const double* V, const double* R0, double* Rho);
/* newton_equations.rs:133-149, :155-166 (the Inverse / LeastSquares arms): V[r] = R^-1 Q^T R[r] with
 * M = QR a Householder factorisation of the symmetric m x m matrix M (row-major; only the lower
 * triangle is read), nrhs 1|2, m <= 16384.  info_out: 0, or k+1 for a zero column / zero R[k][k].
 * ms_out: device time of factorisation + solves. */
int lpipm_k_qr_solve(lpipm_ctx* ctx, uint64_t m, const double* M, int nrhs, const double* R, double* V,
                     int32_t* info_out, double* ms_out);
/* A.w (feasible_point.rs:122, newton_equations.rs:220, residual.rs:23): nrhs (1|2) vectors,
 * W nrhs x n -> Y nrhs x m. */
int lpipm_k_gemv_n(lpipm_ctx* ctx, int nrhs, const double* W, double* Y, int repeats, double* ms_out);
/* A^T.v (feasible_point.rs:123, newton_equations.rs:223, residual.rs:25): V nrhs x m -> U nrhs x n */
int lpipm_k_gemv_t(lpipm_ctx* ctx, int nrhs, const double* V, double* U, int repeats, double* ms_out);
/* One loop body of solve_normal_form (mod.rs:215-222) on the uploaded problem from a GIVEN iterate: get_delta
 * (feasible_point.rs:110-152: residuals, normal equations + factor, Rhat::predictor / corrector, Delta::compute twice,
 * update_gamma), the step length (mod.rs:216-221, feasible_point.rs:53-72) and do_step (:76-106).
 * In/out: x[n], y[m], z[n], *tau, *kappa.  Out: the corrector's direction d_x[n], d_y[m], d_z[n],
 * d_tk = {d_tau, d_kappa}, *alpha, *info (pivot failure as lpipm_k_potrf).  Differential tests of the vector stage. */
int lpipm_k_iteration(lpipm_ctx* ctx, const lpipm_opts* opts, int ip, double* x, double* y, double* z, double* tau,
                      double* kappa, double* d_x, double* d_y, double* d_z, double* d_tk, double* alpha_out,
                      int32_t* info_out);
/* residual.rs:23,25 / feasible_point.rs:122-123 in ONE read of A: Aw_out[m] = A.w, ATv_out[n] = A^T.v (w[n], v[m]). */
int lpipm_k_gemv_dual(lpipm_ctx* ctx, const double* w, const double* v, double* Aw_out, double* ATv_out, int repeats,
                      double* ms_out);
/* ---- InteriorPoint<f32> ------------------------------------------------------------------------
 * src/float.rs:42-43 (`impl Float for f32`): the reference's solver is generic over F; with F = f32 every operation of
 * interior_point/mod.rs:161-168, :199-240 runs in f32.  lpipm_solve_f32 is that instantiation: the same algorithm as
 * lpipm_solve with every operation in f32, on generic (scalar-type-templated) HIP kernels -- correctness first, like the
 * QR arms; the hand-written fp64 path is the fast one.  The slack-form problem (A m x n row-major, lda >= n) is uploaded,
 * solved and released inside the call.  opts: tol and alpha0 are converted to f32; only the Cholesky arm (the default).
 * Return codes as lpipm_solve; x_slack_out[n] = x / tau (also for LPIPM_ITERATION_LIMIT); log nullable, max_iter rows.
 * NOTE (reference behaviour, not a property of this backend): with the default tol = 1e-8 an f32 solve cannot satisfy the
 * optimality test (f32 epsilon is 6e-8): it ends in IterationLimitExceeded or NumericalProblem; pass a tolerance f32 can
 * reach (1e-4 .. 1e-5). */
int lpipm_solve_f32(lpipm_ctx* ctx, uint64_t m, uint64_t n, const float* A_rowmajor, uint64_t lda, const float* b,
                    const float* c, float c0, const lpipm_opts* opts, float* x_slack_out, float* fun_out,
                    uint64_t* iterations_out, lpipm_iter_row_f32* log);
/* Test hook: the SAME generic kernels instantiated for double (solver_generic.hip), so that they can be checked against
 * the fp64 oracle to 1e-8 -- which f32 arithmetic itself cannot show. */
int lpipm_k_generic_solve_f64(lpipm_ctx* ctx, uint64_t m, uint64_t n, const double* A_rowmajor, uint64_t lda, const double* b,
                              const double* c, double c0, const lpipm_opts* opts, double* x_slack_out, double* fun_out,
                              uint64_t* iterations_out, lpipm_iter_row* log);

/* fp64 MFMA issue-rate probe (v_mfma_f64_16x16x4_f64 back to back, operands in registers):
 * tflops_out = achieved TFLOP/s over the whole chip; used to confirm the roofline denominator. */
int lpipm_k_mfma_f64_probe(lpipm_ctx* ctx, int iters, double* tflops_out, double* ms_out);

/* ---- synthetic inputs (SURVEY.md 8d / BASELINE.md 3) -----------------------------------------
 * Equality-form planted LP with a strictly complementary optimum: A_ij ~ N(0,1); basis B of m
 * columns; x*_B ~ U(1,2), x*_N = 0; y* ~ N(0,1); z*_N ~ U(1,2), z*_B = 0; b = A x*, c = A^T y* + z*.
 * splitmix64 -> xoshiro256**, Box-Muller, Fisher-Yates.  Host-side; A m x n row-major.
 * xstar_out (n) is nullable. */
int lpipm_synth_planted_lp(uint64_t seed, uint64_t m, uint64_t n, double* A_out, double* b_out,
                           double* c_out, double* xstar_out);

#ifdef __cplusplus
}
#endif
#endif /* LPIPM_H */
