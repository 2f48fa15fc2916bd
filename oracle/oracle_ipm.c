/*
 * oracle_ipm.c -- line-by-line CPU restatement of the reference's interior-point algorithm
 * (homogeneous self-dual IPM with Mehrotra predictor-corrector), /root/reference/src/solvers/
 * interior_point/{mod,feasible_point,newton_equations,rhat,delta,residual,indicators}.rs.
 * TEST INFRASTRUCTURE ONLY (see oracle_ipm.h).
 *
 * Deliberately keeps the reference's AS-WRITTEN operation counts per iteration:
 * 1 full-square A.(D o A^T) with an n x m temporary, 1 unblocked Cholesky, 4 Cholesky solves
 * (8 triangular sweeps), 6 GEMV-N + 6 GEMV-T, fresh vectors for every intermediate -- so that
 * timing it is a faithful stand-in for `cargo run --release` of the crate (BASELINE.md 2, B-ref).
 * Compiled with -ffp-contract=off: Rust does not contract a*b+c into an fma.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
/* The f32 build of the oracle (liboracle_ipm_f32.so, Makefile): the SAME restatement with every `double` a `float`
 * (and, by -fsingle-precision-constant, every literal too) -- what the reference's generic code is for F = f32
 * (src/float.rs:42-43).  The switch sits behind the system headers: their prototypes must stay what libm implements
 * (sqrt / fabs / fmax of a float argument are exact or correctly rounded after the conversion back). */
#ifdef ORACLE_F32
#define double float
#endif
#include "oracle_ipm.h"

int oracle_qr_factor(uint64_t m, const double* M, double** qr_out, double** beta_out);
int oracle_qr_solve(uint64_t m, const double* QR, const double* beta, const double* b, double* x);

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double* vec(uint64_t n) { return (double*)malloc(sizeof(double) * (n ? n : 1)); }
static double dot(uint64_t n, const double* a, const double* b) {
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

void oracle_default_opts(oracle_opts* o) { /* interior_point/mod.rs:50-60 */
    o->tol = 1e-8;
    o->disp = 0;
    o->ip = 1;
    o->solver_type = 0;
    o->alpha0 = 0.99995;
    o->max_iter = 1000;
}

/* linear_program.rs:125-169 */
int oracle_problem_build(uint64_t n, uint64_t m_ub, const double* A_ub, const double* b_ub,
                         uint64_t m_eq, const double* A_eq, const double* b_eq, const double* c,
                         double* A_out, double* b_out, double* c_out, uint64_t* n_slack_out) {
    if (m_ub + m_eq == 0) return ORACLE_UNCONSTRAINED;           /* :134-136 */
    const uint64_t m = m_ub + m_eq, ns = n + m_ub;
    for (uint64_t i = 0; i < m; ++i) {                            /* :145-156 */
        const double* src = i < m_ub ? A_ub + i * n : A_eq + (i - m_ub) * n;
        for (uint64_t j = 0; j < n; ++j) A_out[i * ns + j] = src[j];
        for (uint64_t j = 0; j < m_ub; ++j) A_out[i * ns + n + j] = (i == j) ? 1.0 : 0.0;
    }
    for (uint64_t i = 0; i < m; ++i) b_out[i] = i < m_ub ? b_ub[i] : b_eq[i - m_ub]; /* :157-158 */
    for (uint64_t j = 0; j < ns; ++j) c_out[j] = j < n ? c[j] : 0.0;                 /* :159-160 */
    *n_slack_out = m_ub;                                                             /* :161 */
    return ORACLE_OK;
}

/* ---- residual.rs:13-44 ------------------------------------------------------------------ */
typedef struct { double rho_p, rho_d, rho_g, rho_mu; } residuals_t;
typedef struct {
    uint64_t m, n;
    const double *A, *b, *c;
    double c0;
    oracle_timing* tm;
} problem_t;

static residuals_t residuals_calculate(const problem_t* P, const double* x, const double* y,
                                       const double* z, double tau, double kappa) {
    const uint64_t m = P->m, n = P->n;
    double t0 = now_s();
    double* Ax = vec(m);
    double* ATy = vec(n);
    oracle_gemv_n(m, n, P->A, x, Ax);  /* residual.rs:23 */
    oracle_gemv_t(m, n, P->A, y, ATy); /* residual.rs:25 */
    if (P->tm) P->tm->gemv += now_s() - t0;
    double sp = 0.0, sd = 0.0;
    for (uint64_t i = 0; i < m; ++i) { double r = P->b[i] * tau - Ax[i]; sp += r * r; }
    for (uint64_t j = 0; j < n; ++j) { double r = P->c[j] * tau - ATy[j] - z[j]; sd += r * r; }
    residuals_t R;
    R.rho_p = sqrt(sp);                                             /* :34 */
    R.rho_d = sqrt(sd);                                             /* :35 */
    R.rho_g = fabs(kappa + dot(n, P->c, x) - dot(m, P->b, y));      /* :27-29,36 */
    R.rho_mu = (dot(n, x, z) + tau * kappa) / (double)(n + 1);      /* :30-32,37 */
    free(Ax);
    free(ATy);
    return R;
}

/* ---- feasible_point.rs:14-21 ------------------------------------------------------------- */
typedef struct {
    double *x, *y, *z;
    double tau, kappa;
    residuals_t initial_residuals;
} point_t;

/* ---- delta.rs:12-18 ---------------------------------------------------------------------- */
typedef struct { double *d_x, *d_y, *d_z; double d_tau, d_kappa; } delta_t;
static void delta_free(delta_t* d) { free(d->d_x); free(d->d_y); free(d->d_z); }

/* ---- rhat.rs:8-14 ------------------------------------------------------------------------ */
typedef struct { double *p, *d, *xs; double g, tk; } rhat_t;
static void rhat_free(rhat_t* r) { free(r->p); free(r->d); free(r->xs); }

/* ---- indicators.rs:8-23 ------------------------------------------------------------------ */
typedef struct { double rho_p, rho_d, rho_A, rho_g, rho_mu, obj, bty; } indicators_t;
enum { ST_OPTIMAL, ST_INFEASIBLE, ST_UNBOUNDED, ST_UNFINISHED };

static indicators_t indicators_from_point(const point_t* pt, const problem_t* P) {
    const uint64_t m = P->m, n = P->n;
    indicators_t I;
    double obj = 0.0;
    for (uint64_t j = 0; j < n; ++j) obj += P->c[j] * (pt->x[j] / pt->tau); /* indicators.rs:41 */
    I.obj = obj + P->c0;
    I.bty = dot(m, P->b, pt->y);                                            /* :42 */
    I.rho_A = fabs(dot(n, P->c, pt->x) - I.bty) / (pt->tau + fabs(dot(m, P->b, pt->y))); /* :43-44 */
    residuals_t r = residuals_calculate(P, pt->x, pt->y, pt->z, pt->tau, pt->kappa);     /* :45 */
    const residuals_t* r0 = &pt->initial_residuals;
    I.rho_p = r.rho_p / fmax(r0->rho_p, 1.0);  /* :47 */
    I.rho_d = r.rho_d / fmax(r0->rho_d, 1.0);  /* :48 */
    I.rho_g = r.rho_g / fmax(r0->rho_g, 1.0);  /* :50 */
    I.rho_mu = r.rho_mu / r0->rho_mu;          /* :51 */
    return I;
}

static int indicators_status(const indicators_t* I, double tau, double kappa, double tol) {
    int tau_too_small = tau < tol * fmax(kappa, 1.0);                               /* :67 */
    int inf1 = (I->rho_p < tol && I->rho_d < tol && I->rho_g < tol) && tau_too_small; /* :63,68 */
    int inf2 = I->rho_mu < tol && tau_too_small;                                     /* :69 */
    if (inf1 || inf2) return I->bty > tol ? ST_INFEASIBLE : ST_UNBOUNDED;            /* :70-76 */
    if (I->rho_p < tol && I->rho_d < tol && I->rho_A < tol) return ST_OPTIMAL;       /* :58,77-79 */
    return ST_UNFINISHED;
}

/* ---- newton_equations.rs ----------------------------------------------------------------- */
typedef struct {
    int kind; /* 0 Cholesky, 1 Inv (QR), 2 LstSq (QR) */
    uint64_t m;
    double* M;      /* kept for the fallback chain, :202-207 */
    double* factor; /* Cholesky L, or packed QR */
    double* beta;   /* QR only */
    double* Dinv;
} eqsolver_t;

static void eqsolver_free(eqsolver_t* s) { free(s->M); free(s->factor); free(s->beta); free(s->Dinv); }

static int eqsolver_factor(eqsolver_t* s, int kind, oracle_timing* tm) {
    double t0 = now_s();
    int rc = 0;
    free(s->factor); s->factor = NULL;
    free(s->beta); s->beta = NULL;
    s->kind = kind;
    if (kind == 0) { /* :129-131 */
        s->factor = vec(s->m * s->m);
        memcpy(s->factor, s->M, sizeof(double) * s->m * s->m);
        rc = oracle_cholesky(s->m, s->factor);
    } else {         /* :133-149 */
        rc = oracle_qr_factor(s->m, s->M, &s->factor, &s->beta);
    }
    if (tm) tm->chol += now_s() - t0;
    return rc;
}

/* EquationSolverType::build, newton_equations.rs:48-64 */
static int eqsolver_build(eqsolver_t* s, int solver_type, const point_t* pt, const problem_t* P) {
    const uint64_t m = P->m, n = P->n;
    memset(s, 0, sizeof(*s));
    s->m = m;
    s->Dinv = vec(n);
    for (uint64_t j = 0; j < n; ++j) s->Dinv[j] = pt->x[j] / pt->z[j]; /* :54 */
    s->M = vec(m * m);
    double t0 = now_s();
    oracle_adat(m, n, P->A, s->Dinv, s->M);                            /* :55-57 */
    if (P->tm) P->tm->adat += now_s() - t0;
    if (eqsolver_factor(s, solver_type, P->tm)) return ORACLE_NUMERICAL_PROBLEM; /* :58-63 */
    return ORACLE_OK;
}

/* EquationsSolver::solve, :151-169 */
static int eqsolver_solve(eqsolver_t* s, const double* r, double* v, oracle_timing* tm) {
    double t0 = now_s();
    int rc = 0;
    if (s->kind == 0) oracle_cholesky_solve(s->m, s->factor, r, v);
    else rc = oracle_qr_solve(s->m, s->factor, s->beta, r, v);
    if (tm) tm->solves += now_s() - t0;
    return rc;
}

/* sym_solve, :214-225 ([1] eq. 8.31 / 8.32).  u[n], v[m] freshly allocated by the caller. */
static int sym_solve(eqsolver_t* s, const problem_t* P, const double* r1, const double* r2, double* u,
                     double* v) {
    const uint64_t m = P->m, n = P->n;
    double* w = vec(n);
    double* r = vec(m);
    double* t = vec(n);
    for (uint64_t j = 0; j < n; ++j) w[j] = s->Dinv[j] * r1[j];
    double t0 = now_s();
    oracle_gemv_n(m, n, P->A, w, r);
    if (P->tm) P->tm->gemv += now_s() - t0;
    for (uint64_t i = 0; i < m; ++i) r[i] = r2[i] + r[i];              /* :220 */
    int rc = eqsolver_solve(s, r, v, P->tm);                           /* :221 */
    if (!rc) {
        t0 = now_s();
        oracle_gemv_t(m, n, P->A, v, t);
        if (P->tm) P->tm->gemv += now_s() - t0;
        for (uint64_t j = 0; j < n; ++j) u[j] = s->Dinv[j] * (t[j] - r1[j]); /* :223 */
    }
    free(w); free(r); free(t);
    return rc;
}

typedef struct { double *p, *q, *u, *v; } newton_t;
static void newton_free(newton_t* N) { free(N->p); free(N->q); free(N->u); free(N->v); }

/* solve_newton_equations, :176-210 */
static int solve_newton_equations(eqsolver_t* s, const problem_t* P, const double* x,
                                  const rhat_t* rhat, newton_t* out) {
    const uint64_t m = P->m, n = P->n;
    for (;;) {
        out->p = vec(n); out->q = vec(m); out->u = vec(n); out->v = vec(m);
        double* r1 = vec(n);
        for (uint64_t j = 0; j < n; ++j) r1[j] = rhat->d[j] - rhat->xs[j] / x[j]; /* :188 */
        int rc1 = sym_solve(s, P, P->c, P->b, out->p, out->q);                    /* :187 */
        int rc2 = sym_solve(s, P, r1, rhat->p, out->u, out->v);                   /* :188 */
        free(r1);
        if (!rc1 && !rc2) {
            int nan = 0;                                                          /* :190-194 */
            for (uint64_t j = 0; j < n; ++j) nan |= (out->p[j] != out->p[j]);
            for (uint64_t i = 0; i < m; ++i) nan |= (out->q[i] != out->q[i]);
            if (nan) { newton_free(out); return ORACLE_NUMERICAL_PROBLEM; }
            return ORACLE_OK;
        }
        newton_free(out);
        if (s->kind == 2) return ORACLE_NUMERICAL_PROBLEM;                        /* :208 */
        if (eqsolver_factor(s, s->kind + 1, P->tm)) return ORACLE_NUMERICAL_PROBLEM; /* :202-207 */
    }
}

/* Delta::compute, delta.rs:21-49 */
static int delta_compute(const point_t* pt, const rhat_t* rhat, const problem_t* P, eqsolver_t* s,
                         delta_t* D) {
    const uint64_t m = P->m, n = P->n;
    newton_t N;
    int rc = solve_newton_equations(s, P, pt->x, rhat, &N);                      /* :27 */
    if (rc) return rc;
    double t0 = now_s();
    /* :29-32 */
    D->d_tau = (rhat->g + 1.0 / pt->tau * rhat->tk - (-dot(n, P->c, N.u) + dot(m, P->b, N.v))) /
               (1.0 / pt->tau * pt->kappa + (-dot(n, P->c, N.p) + dot(m, P->b, N.q)));
    D->d_x = vec(n); D->d_y = vec(m); D->d_z = vec(n);
    for (uint64_t j = 0; j < n; ++j) D->d_x[j] = N.u[j] + N.p[j] * D->d_tau;     /* :33 */
    for (uint64_t i = 0; i < m; ++i) D->d_y[i] = N.v[i] + N.q[i] * D->d_tau;     /* :34 */
    for (uint64_t j = 0; j < n; ++j)
        D->d_z[j] = (rhat->xs[j] - pt->z[j] * D->d_x[j]) / pt->x[j];             /* :37 */
    D->d_kappa = 1.0 / pt->tau * (rhat->tk - pt->kappa * D->d_tau);              /* :38 */
    newton_free(&N);
    if (P->tm) P->tm->rest += now_s() - t0;
    return ORACLE_OK;
}

/* Rhat::predictor, rhat.rs:17-35 */
static rhat_t rhat_predictor(const problem_t* P, const double* r_P, const double* r_D, double r_G,
                             double eta, const point_t* pt, double gamma, double mu) {
    const uint64_t m = P->m, n = P->n;
    rhat_t R;
    R.p = vec(m); R.d = vec(n); R.xs = vec(n);
    for (uint64_t i = 0; i < m; ++i) R.p[i] = r_P[i] * eta;
    for (uint64_t j = 0; j < n; ++j) R.d[j] = r_D[j] * eta;
    R.g = r_G * eta;
    for (uint64_t j = 0; j < n; ++j) R.xs[j] = (pt->x[j] * -1.0) * pt->z[j] + gamma * mu; /* :32 */
    R.tk = gamma * mu - pt->tau * pt->kappa;                                               /* :33 */
    return R;
}

/* Rhat::corrector, rhat.rs:37-75 */
static rhat_t rhat_corrector(const problem_t* P, const double* r_P, const double* r_D, double r_G,
                             double eta, const point_t* pt, const delta_t* dl, double gamma,
                             double mu, double alpha, int ip) {
    const uint64_t m = P->m, n = P->n;
    rhat_t R;
    R.p = vec(m); R.d = vec(n); R.xs = vec(n);
    if (ip) { /* eq. 8.23, :51-60 */
        double alpha_2 = alpha * alpha;
        for (uint64_t j = 0; j < n; ++j)
            R.xs[j] = (pt->x[j] * -1.0) * pt->z[j] - (dl->d_x[j] * dl->d_z[j]) * alpha_2 +
                      (1.0 - alpha) * gamma * mu;
        R.tk = (1.0 - alpha) * gamma * mu - pt->tau * pt->kappa - alpha_2 * dl->d_tau * dl->d_kappa;
    } else {  /* eq. 8.13, :62-66 */
        for (uint64_t j = 0; j < n; ++j)
            R.xs[j] = (pt->x[j] * -1.0) * pt->z[j] + gamma * mu - (dl->d_x[j] * dl->d_z[j]);
        R.tk = gamma * mu - pt->tau * pt->kappa - dl->d_tau * dl->d_kappa;
    }
    for (uint64_t i = 0; i < m; ++i) R.p[i] = r_P[i] * eta;
    for (uint64_t j = 0; j < n; ++j) R.d[j] = r_D[j] * eta;
    R.g = r_G * eta;
    return R;
}

/* get_step_size, feasible_point.rs:53-72 */
static double min_ratio(double deflt, double d_x, double x) {
    if (d_x < 0.0) return fmin(deflt, x / -d_x);
    return deflt;
}
static double get_step_size(const point_t* pt, const delta_t* D, uint64_t n, double alpha0) {
    double alpha_x = 1.0, alpha_z = 1.0;
    for (uint64_t j = 0; j < n; ++j) alpha_x = min_ratio(alpha_x, D->d_x[j], pt->x[j]);
    for (uint64_t j = 0; j < n; ++j) alpha_z = min_ratio(alpha_z, D->d_z[j], pt->z[j]);
    double alpha_tau = min_ratio(1.0, D->d_tau, pt->tau);
    double alpha_kappa = min_ratio(1.0, D->d_kappa, pt->kappa);
    return fmin(fmin(fmin(fmin(1.0, alpha_x), alpha_tau), alpha_z), alpha_kappa) * alpha0; /* :66-71 */
}

/* update_gamma, feasible_point.rs:156-165 */
static double update_gamma(int ip, double alpha) {
    if (ip) return 10.0;
    double beta1 = 0.1;
    return (1.0 - alpha) * (1.0 - alpha) * fmin(beta1, 1.0 - alpha);
}

/* get_delta, feasible_point.rs:110-152 */
static int get_delta(const point_t* pt, const problem_t* P, int solver_type, int ip, delta_t* out) {
    const uint64_t m = P->m, n = P->n;
    double gamma = ip ? 1.0 : 0.0;           /* :119 */
    double eta = ip ? 1.0 : 1.0 - gamma;     /* :120 */
    double* r_P = vec(m);
    double* r_D = vec(n);
    double t0 = now_s();
    oracle_gemv_n(m, n, P->A, pt->x, r_P);
    oracle_gemv_t(m, n, P->A, pt->y, r_D);
    if (P->tm) P->tm->gemv += now_s() - t0;
    for (uint64_t i = 0; i < m; ++i) r_P[i] = P->b[i] * pt->tau - r_P[i];             /* :122 */
    for (uint64_t j = 0; j < n; ++j) r_D[j] = P->c[j] * pt->tau - r_D[j] - pt->z[j];  /* :123 */
    double r_G = dot(n, P->c, pt->x) - dot(m, P->b, pt->y) + pt->kappa;                /* :124 */
    double mu = (dot(n, pt->x, pt->z) + pt->tau * pt->kappa) / (double)(n + 1);        /* :125 */

    eqsolver_t S;
    int rc = eqsolver_build(&S, solver_type, pt, P);                                   /* :127 */
    if (rc) { eqsolver_free(&S); free(r_P); free(r_D); return rc; }

    rhat_t rh = rhat_predictor(P, r_P, r_D, r_G, eta, pt, gamma, mu);                  /* :129 */
    delta_t pred;
    rc = delta_compute(pt, &rh, P, &S, &pred);                                         /* :130-131 */
    rhat_free(&rh);
    if (rc) { eqsolver_free(&S); free(r_P); free(r_D); return rc; }

    double alpha = get_step_size(pt, &pred, n, 1.0);                                   /* :134 */
    gamma = update_gamma(ip, alpha);                                                   /* :135 */
    eta = ip ? 1.0 : 1.0 - gamma;                                                      /* :136 */
    rh = rhat_corrector(P, r_P, r_D, r_G, eta, pt, &pred, gamma, mu, alpha, ip);       /* :137-148 */
    rc = delta_compute(pt, &rh, P, &S, out);                                           /* :149 */
    rhat_free(&rh);
    delta_free(&pred);
    eqsolver_free(&S);
    free(r_P); free(r_D);
    return rc;
}

/* do_step, feasible_point.rs:76-106 */
static void do_step(point_t* pt, const delta_t* D, double alpha, int ip, uint64_t m, uint64_t n) {
    for (uint64_t j = 0; j < n; ++j) pt->x[j] = pt->x[j] + D->d_x[j] * alpha;
    for (uint64_t i = 0; i < m; ++i) pt->y[i] = pt->y[i] + D->d_y[i] * alpha;
    for (uint64_t j = 0; j < n; ++j) pt->z[j] = pt->z[j] + D->d_z[j] * alpha;
    pt->tau = pt->tau + D->d_tau * alpha;
    pt->kappa = pt->kappa + D->d_kappa * alpha;
    if (ip) { /* :87-95 */
        for (uint64_t j = 0; j < n; ++j) pt->x[j] = fmax(pt->x[j], 1.0);
        for (uint64_t j = 0; j < n; ++j) pt->z[j] = fmax(pt->z[j], 1.0);
        pt->tau = fmax(pt->tau, 1.0);
        pt->kappa = fmax(pt->kappa, 1.0);
    }
}

/* ONE pass of the loop body of solve_normal_form (mod.rs:215-222) from a GIVEN iterate: get_delta
 * (feasible_point.rs:110-152), the step length (mod.rs:216-221), do_step (feasible_point.rs:76-106).  For the
 * differential tests of the device's vector stage (rhat.rs, delta.rs, the ratio test, the step) on arbitrary
 * iterates -- including ip = 1 and directions with zero / negative entries -- not only along trajectories.
 * In/out: x[n], y[m], z[n], *tau, *kappa.  Out: d_x[n], d_y[m], d_z[n], d_tk[2] = {d_tau, d_kappa}, *alpha. */
int oracle_iteration(uint64_t m, uint64_t n, const double* A, const double* b, const double* c, int solver_type,
                     int ip, double alpha0, double* x, double* y, double* z, double* tau, double* kappa,
                     double* d_x, double* d_y, double* d_z, double* d_tk, double* alpha_out) {
    problem_t P = {m, n, A, b, c, 0.0, NULL};
    point_t pt;
    pt.x = x; pt.y = y; pt.z = z; pt.tau = *tau; pt.kappa = *kappa;
    memset(&pt.initial_residuals, 0, sizeof(pt.initial_residuals));
    delta_t D;
    int rc = get_delta(&pt, &P, solver_type, ip, &D);
    if (rc) return rc;
    const double alpha = ip ? 1.0 : get_step_size(&pt, &D, n, alpha0);
    memcpy(d_x, D.d_x, sizeof(double) * n); memcpy(d_y, D.d_y, sizeof(double) * m); memcpy(d_z, D.d_z, sizeof(double) * n);
    d_tk[0] = D.d_tau; d_tk[1] = D.d_kappa;
    do_step(&pt, &D, alpha, ip, m, n);
    *tau = pt.tau; *kappa = pt.kappa; *alpha_out = alpha;
    delta_free(&D);
    return ORACLE_OK;
}

/* InteriorPoint::solve_normal_form + solve, interior_point/mod.rs:199-240, :161-168 */
int oracle_ipm_solve(uint64_t m, uint64_t n, const double* A, const double* b, const double* c,
                     double c0, const oracle_opts* opts, double* x_slack_out, double* fun_out,
                     uint64_t* iterations_out, oracle_iter_row* log, oracle_timing* timing) {
    /* InteriorPointBuilder::build validation, mod.rs:118-128 */
    if (!(opts->alpha0 > 0.0) || !(opts->alpha0 < 1.0)) return ORACLE_INVALID_PARAMETER;
    if (!(opts->tol > 0.0)) return ORACLE_INVALID_PARAMETER;
    if (opts->solver_type < 0 || opts->solver_type > 2) return ORACLE_INVALID_PARAMETER;
    if (m == 0) return ORACLE_UNCONSTRAINED;

    oracle_timing tm_local;
    memset(&tm_local, 0, sizeof(tm_local));
    problem_t P = {m, n, A, b, c, c0, &tm_local};
    double t_start = now_s();

    /* FeasiblePoint::blind_start, feasible_point.rs:24-39 */
    point_t pt;
    pt.x = vec(n); pt.y = vec(m); pt.z = vec(n);
    for (uint64_t j = 0; j < n; ++j) pt.x[j] = 1.0;
    for (uint64_t i = 0; i < m; ++i) pt.y[i] = 0.0;
    for (uint64_t j = 0; j < n; ++j) pt.z[j] = 1.0;
    pt.tau = 1.0;
    pt.kappa = 1.0;
    pt.initial_residuals = residuals_calculate(&P, pt.x, pt.y, pt.z, pt.tau, pt.kappa);

    indicators_t ind = indicators_from_point(&pt, &P); /* mod.rs:206 */
    if (opts->disp) {                                  /* mod.rs:208-211 */
        printf("alpha     \trho_p     \trho_d     \trho_g     \trho_mu    \tobj       \n");
        printf("1.00000000\t%.8f\t%.8f\t%.8f\t%.8f\t%8.3f\n", ind.rho_p, ind.rho_d, ind.rho_g,
               ind.rho_mu, ind.obj);
    }
    int ip = opts->ip;
    int ret = ORACLE_ITERATION_LIMIT;
    uint64_t iteration = 0;
    for (iteration = 1; iteration <= opts->max_iter; ++iteration) { /* mod.rs:213 */
        delta_t D;
        int rc = get_delta(&pt, &P, opts->solver_type, ip, &D);     /* :215 */
        if (rc) { ret = rc; break; }
        double alpha = ip ? 1.0 : get_step_size(&pt, &D, n, opts->alpha0); /* :216-221 */
        double t0 = now_s();
        do_step(&pt, &D, alpha, ip, m, n);                          /* :222 */
        tm_local.rest += now_s() - t0;
        delta_free(&D);
        ip = 0;                                                     /* :223 */
        ind = indicators_from_point(&pt, &P);                       /* :225 */
        if (opts->disp)
            printf("%.8f\t%.8f\t%.8f\t%.8f\t%.8f\t%8.3f\n", alpha, ind.rho_p, ind.rho_d, ind.rho_g,
                   ind.rho_mu, ind.obj);
        if (log) {
            oracle_iter_row* r = &log[iteration - 1];
            r->alpha = alpha; r->rho_p = ind.rho_p; r->rho_d = ind.rho_d; r->rho_A = ind.rho_A;
            r->rho_g = ind.rho_g; r->rho_mu = ind.rho_mu; r->obj = ind.obj;
        }
        int st = indicators_status(&ind, pt.tau, pt.kappa, opts->tol); /* :230 */
        if (st == ST_OPTIMAL) { ret = ORACLE_OK; break; }
        if (st == ST_INFEASIBLE) { ret = ORACLE_INFEASIBLE; break; }
        if (st == ST_UNBOUNDED) { ret = ORACLE_UNBOUNDED; break; }
    }
    if (ret == ORACLE_ITERATION_LIMIT) iteration = opts->max_iter;
    if (ret == ORACLE_OK || ret == ORACLE_ITERATION_LIMIT) {
        /* mod.rs:231 / :237-239: x / tau; mod.rs:165: fun = c.x + c0 */
        for (uint64_t j = 0; j < n; ++j) x_slack_out[j] = pt.x[j] / pt.tau;
        if (fun_out) *fun_out = dot(n, c, x_slack_out) + c0;
    }
    if (iterations_out) *iterations_out = iteration;
    tm_local.total = now_s() - t_start;
    if (timing) *timing = tm_local;
    free(pt.x); free(pt.y); free(pt.z);
    return ret;
}
