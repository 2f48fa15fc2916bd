/*
 * oracle_linalg.c -- the dense fp64 arithmetic the reference delegates to un-vendored crates
 * (ndarray 0.15 + matrixmultiply, linfa-linalg 0.1; /root/reference/Cargo.toml:31-36).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_ipm.h).  Single-threaded on purpose: the reference's
 * default backend has no rayon/threading feature enabled (Cargo.toml:31-33).
 *
 * The crates' sources are not in /root/reference, so their published algorithms are restated:
 *   - ndarray `.dot` on two 2-D f64 arrays -> matrixmultiply::dgemm, a packed, cache-blocked
 *     GEMM with a register-tiled FMA micro-kernel (restated below as gemm_nn: MC/KC/NC packing
 *     + a 6x8 AVX2 micro-kernel).
 *   - ndarray `.dot` matrix x vector -> row dot products (C-contiguous lhs) / axpy sweep
 *     (transposed view).
 *   - linfa_linalg::cholesky::Cholesky::cholesky -> unblocked row-by-row (Cholesky-Banachiewicz)
 *     factorisation returning the LOWER factor, error on a non-positive pivot.
 *   - linfa_linalg::cholesky::SolveCInplace::solvec_into -> forward then transposed-backward
 *     substitution with that factor.
 *   - linfa_linalg::qr -> Householder QR; solve_into = R^-1 Q^T b.
 * Call sites in the reference: newton_equations.rs:55-57,130,134,139-141,154-164;
 * feasible_point.rs:122-125; residual.rs:22-31; indicators.rs:41-44; delta.rs:30-32.
 * Any correct fp64 kernel is admissible here: the reference itself ships two interchangeable
 * backends judged at 1e-6 on x (.github/workflows/testing.yml:33,58).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
/* The f32 build of the oracle (liboracle_ipm_f32.so, Makefile): the SAME restatement with every `double` a `float`
 * (and, by -fsingle-precision-constant, every literal too) -- what the reference's generic code is for F = f32
 * (src/float.rs:42-43).  The switch sits behind the system headers: their prototypes must stay what libm implements
 * (sqrt / fabs / fmax of a float argument are exact or correctly rounded after the conversion back). */
#ifdef ORACLE_F32
#define double float
#endif
#include "oracle_ipm.h"

#ifndef ORACLE_F32
typedef double v4d __attribute__((vector_size(32)));
#endif

/* ------------------------------------------------------------------ GEMM C = A(m x k) . B(k x n) */
#define MR 6
#define NR 8
#define MC 96
#define KC 256
#define NC 2048

static void pack_a(int mc, int kc, const double* A, int lda, double* Ap) {
    /* panels of MR rows, column-by-column inside a panel, zero padded */
    for (int i0 = 0; i0 < mc; i0 += MR) {
        int mr = mc - i0 < MR ? mc - i0 : MR;
        for (int p = 0; p < kc; ++p) {
            for (int i = 0; i < mr; ++i) Ap[i] = A[(size_t)(i0 + i) * lda + p];
            for (int i = mr; i < MR; ++i) Ap[i] = 0.0;
            Ap += MR;
        }
    }
}
static void pack_b(int kc, int nc, const double* B, int ldb, double* Bp) {
    for (int j0 = 0; j0 < nc; j0 += NR) {
        int nr = nc - j0 < NR ? nc - j0 : NR;
        for (int p = 0; p < kc; ++p) {
            const double* src = B + (size_t)p * ldb + j0;
            for (int j = 0; j < nr; ++j) Bp[j] = src[j];
            for (int j = nr; j < NR; ++j) Bp[j] = 0.0;
            Bp += NR;
        }
    }
}
#ifdef ORACLE_F32
/* acc[MR][NR] = Ap(MR x kc) . Bp(kc x NR): the f32 build's micro-kernel, element by element (the vector form below is laid
 * out for 8-byte elements); the same sums in the same k order */
static inline void micro(int kc, const double* Ap, const double* Bp, double* acc /*MR*NR*/) {
    for (int i = 0; i < MR * NR; ++i) acc[i] = 0.0;
    for (int p = 0; p < kc; ++p) {
        for (int i = 0; i < MR; ++i)
            for (int j = 0; j < NR; ++j) acc[i * NR + j] += Ap[i] * Bp[j];
        Ap += MR;
        Bp += NR;
    }
}
#else
/* acc[MR][NR] += Ap(MR x kc) . Bp(kc x NR) */
static inline void micro(int kc, const double* Ap, const double* Bp, double* acc /*MR*NR*/) {
    v4d c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0}, c20 = {0}, c21 = {0};
    v4d c30 = {0}, c31 = {0}, c40 = {0}, c41 = {0}, c50 = {0}, c51 = {0};
    for (int p = 0; p < kc; ++p) {
        v4d b0, b1;
        memcpy(&b0, Bp, 32);
        memcpy(&b1, Bp + 4, 32);
        v4d a;
        a = (v4d){Ap[0], Ap[0], Ap[0], Ap[0]}; c00 += a * b0; c01 += a * b1;
        a = (v4d){Ap[1], Ap[1], Ap[1], Ap[1]}; c10 += a * b0; c11 += a * b1;
        a = (v4d){Ap[2], Ap[2], Ap[2], Ap[2]}; c20 += a * b0; c21 += a * b1;
        a = (v4d){Ap[3], Ap[3], Ap[3], Ap[3]}; c30 += a * b0; c31 += a * b1;
        a = (v4d){Ap[4], Ap[4], Ap[4], Ap[4]}; c40 += a * b0; c41 += a * b1;
        a = (v4d){Ap[5], Ap[5], Ap[5], Ap[5]}; c50 += a * b0; c51 += a * b1;
        Ap += MR;
        Bp += NR;
    }
    memcpy(acc + 0, &c00, 32);  memcpy(acc + 4, &c01, 32);
    memcpy(acc + 8, &c10, 32);  memcpy(acc + 12, &c11, 32);
    memcpy(acc + 16, &c20, 32); memcpy(acc + 20, &c21, 32);
    memcpy(acc + 24, &c30, 32); memcpy(acc + 28, &c31, 32);
    memcpy(acc + 32, &c40, 32); memcpy(acc + 36, &c41, 32);
    memcpy(acc + 40, &c50, 32); memcpy(acc + 44, &c51, 32);
}

#endif

/* C(m x n, ldc) = A(m x k, lda) . B(k x n, ldb), all row-major */
static void gemm_nn(int m, int n, int k, const double* A, int lda, const double* B, int ldb,
                    double* C, int ldc) {
    for (int i = 0; i < m; ++i) memset(C + (size_t)i * ldc, 0, sizeof(double) * (size_t)n);
    double* Ap = (double*)aligned_alloc(64, sizeof(double) * (size_t)(MC + MR) * KC);
    double* Bp = (double*)aligned_alloc(64, sizeof(double) * (size_t)KC * (NC + NR));
    double acc[MR * NR];
    for (int jc = 0; jc < n; jc += NC) {
        int nc = n - jc < NC ? n - jc : NC;
        for (int pc = 0; pc < k; pc += KC) {
            int kc = k - pc < KC ? k - pc : KC;
            pack_b(kc, nc, B + (size_t)pc * ldb + jc, ldb, Bp);
            for (int ic = 0; ic < m; ic += MC) {
                int mc = m - ic < MC ? m - ic : MC;
                pack_a(mc, kc, A + (size_t)ic * lda + pc, lda, Ap);
                for (int jr = 0; jr < nc; jr += NR) {
                    int nr = nc - jr < NR ? nc - jr : NR;
                    for (int ir = 0; ir < mc; ir += MR) {
                        int mr = mc - ir < MR ? mc - ir : MR;
                        micro(kc, Ap + (size_t)(ir / MR) * MR * kc, Bp + (size_t)(jr / NR) * NR * kc,
                              acc);
                        double* c = C + (size_t)(ic + ir) * ldc + jc + jr;
                        for (int i = 0; i < mr; ++i)
                            for (int j = 0; j < nr; ++j) c[(size_t)i * ldc + j] += acc[i * NR + j];
                    }
                }
            }
        }
    }
    free(Ap);
    free(Bp);
}

/* newton_equations.rs:54-57
 *   let M = problem.A().dot(&(&Dinv.clone().insert_axis(Axis(1)) * &problem.A().t()));
 * i.e. an n x m temporary T[k][j] = Dinv[k] * A[j][k], then the full-square product A . T. */
void oracle_adat(uint64_t m, uint64_t n, const double* A, const double* dinv, double* M) {
    double* T = (double*)malloc(sizeof(double) * m * n);
    for (uint64_t k = 0; k < n; ++k)
        for (uint64_t j = 0; j < m; ++j) T[k * m + j] = dinv[k] * A[j * n + k];
    gemm_nn((int)m, (int)m, (int)n, A, (int)n, T, (int)m, M, (int)m);
    free(T);
}

/* newton_equations.rs:129-131  `M.cholesky()` (linfa-linalg): unblocked, lower factor. */
int oracle_cholesky(uint64_t m, double* M) {
    for (uint64_t j = 0; j < m; ++j) {
        double* Lj = M + j * m;
        double d = 0.0;
        for (uint64_t k = 0; k < j; ++k) {
            const double* Lk = M + k * m;
            double s = 0.0;
            for (uint64_t i = 0; i < k; ++i) s += Lk[i] * Lj[i];
            s = (Lj[k] - s) / Lk[k];
            Lj[k] = s;
            d += s * s;
        }
        d = Lj[j] - d;
        if (!(d > 0.0)) return (int)(j + 1);
        Lj[j] = sqrt(d);
        for (uint64_t k = j + 1; k < m; ++k) Lj[k] = 0.0;
    }
    return 0;
}

/* newton_equations.rs:151-169  `factor.solvec_into(b2)`: L w = r, then L^T v = w. */
void oracle_cholesky_solve(uint64_t m, const double* L, const double* r, double* v) {
    for (uint64_t i = 0; i < m; ++i) {
        const double* Li = L + i * m;
        double s = r[i];
        for (uint64_t k = 0; k < i; ++k) s -= Li[k] * v[k];
        v[i] = s / Li[i];
    }
    for (uint64_t ii = m; ii-- > 0;) {
        double s = v[ii] / L[ii * m + ii];
        v[ii] = s;
        const double* Li = L + ii * m;
        for (uint64_t k = 0; k < ii; ++k) v[k] -= Li[k] * s;
    }
}

/* `A.dot(&w)` for C-contiguous A: one dot product per row (feasible_point.rs:122, residual.rs:23,
 * newton_equations.rs:220) */
void oracle_gemv_n(uint64_t m, uint64_t n, const double* A, const double* w, double* y) {
    for (uint64_t i = 0; i < m; ++i) {
        const double* Ai = A + i * n;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        uint64_t k = 0;
        for (; k + 4 <= n; k += 4) {
            s0 += Ai[k] * w[k];
            s1 += Ai[k + 1] * w[k + 1];
            s2 += Ai[k + 2] * w[k + 2];
            s3 += Ai[k + 3] * w[k + 3];
        }
        for (; k < n; ++k) s0 += Ai[k] * w[k];
        y[i] = (s0 + s1) + (s2 + s3);
    }
}

/* `A.t().dot(&v)` (feasible_point.rs:123, residual.rs:25, newton_equations.rs:223) */
void oracle_gemv_t(uint64_t m, uint64_t n, const double* A, const double* v, double* u) {
    for (uint64_t k = 0; k < n; ++k) u[k] = 0.0;
    for (uint64_t i = 0; i < m; ++i) {
        const double* Ai = A + i * n;
        const double vi = v[i];
        for (uint64_t k = 0; k < n; ++k) u[k] += vi * Ai[k];
    }
}

/* ------------------------------------------------------------------ Householder QR (Inverse / LeastSquares arms)
 * newton_equations.rs:133-149 (`M.qr()`), :155-166 (`solve_into` / `solve_tr_into`).  M is square
 * here, so the LeastSquares arm takes the `nrows >= ncols` branch (:138-139) and is the same
 * QR solve as the Inverse arm. */
typedef struct { uint64_t m; double* QR; double* beta; } oracle_qr;

int oracle_qr_factor(uint64_t m, const double* M, double** qr_out, double** beta_out) {
    double* R = (double*)malloc(sizeof(double) * m * m);
    double* beta = (double*)malloc(sizeof(double) * m);
    memcpy(R, M, sizeof(double) * m * m);
    for (uint64_t k = 0; k < m; ++k) {
        double nrm = 0.0;
        for (uint64_t i = k; i < m; ++i) nrm += R[i * m + k] * R[i * m + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0 || nrm != nrm) { free(R); free(beta); return 1; }
        double akk = R[k * m + k];
        double alpha = akk > 0 ? -nrm : nrm;
        double v0 = akk - alpha;
        /* v = [v0, R[k+1:,k]] ; beta = 2 / (v.v) */
        double vv = v0 * v0;
        for (uint64_t i = k + 1; i < m; ++i) vv += R[i * m + k] * R[i * m + k];
        beta[k] = vv == 0.0 ? 0.0 : 2.0 / vv;
        for (uint64_t j = k + 1; j < m; ++j) {
            double s = v0 * R[k * m + j];
            for (uint64_t i = k + 1; i < m; ++i) s += R[i * m + k] * R[i * m + j];
            s *= beta[k];
            R[k * m + j] -= s * v0;
            for (uint64_t i = k + 1; i < m; ++i) R[i * m + j] -= s * R[i * m + k];
        }
        R[k * m + k] = alpha;
        /* store v below the diagonal scaled so that v0 = 1 */
        for (uint64_t i = k + 1; i < m; ++i) R[i * m + k] /= v0;
        beta[k] *= v0 * v0;
    }
    *qr_out = R;
    *beta_out = beta;
    return 0;
}

int oracle_qr_solve(uint64_t m, const double* QR, const double* beta, const double* b, double* x) {
    memcpy(x, b, sizeof(double) * m);
    for (uint64_t k = 0; k < m; ++k) { /* x = Q^T b */
        double s = x[k];
        for (uint64_t i = k + 1; i < m; ++i) s += QR[i * m + k] * x[i];
        s *= beta[k];
        x[k] -= s;
        for (uint64_t i = k + 1; i < m; ++i) x[i] -= s * QR[i * m + k];
    }
    for (uint64_t ii = m; ii-- > 0;) { /* R x = Q^T b */
        double s = x[ii];
        for (uint64_t j = ii + 1; j < m; ++j) s -= QR[ii * m + j] * x[j];
        double r = QR[ii * m + ii];
        if (r == 0.0) return 1;
        x[ii] = s / r;
    }
    return 0;
}
