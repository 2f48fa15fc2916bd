// solver_generic.hip -- the interior-point loop once more, GENERIC IN THE SCALAR TYPE: what `InteriorPoint<F>::solve` is for
// F = f32 (reference src/float.rs:42-43 `impl Float for f32`).
//
// The product's fast path (solver.hip + kernels_*.hip) is hand-written fp64 -- the BASELINE metric -- and cannot be
// re-instantiated: the MFMA fragment layouts, the DPP pivot chain and the LDS images are fp64-specific.  The reference's
// f32 instantiation is the same algorithm with every operation in f32, and that is what this file provides: the hot path
// of interior_point/{mod,feasible_point,newton_equations,rhat,delta,residual,indicators}.rs as plain HIP kernels templated
// on T, correctness first (like the QR arms, kernels_qr.hip): LDS-tiled FMA GEMM for A.D.A^T and the factorisation's
// updates, a blocked right-looking Cholesky (64-blocks, diagonal block and its inverse by one workgroup), block
// substitution for the solves, row / column GEMVs, and the vector stage as single-workgroup kernels whose reductions are
// fixed-order trees.  No CPU fallback: everything numerical below is a kernel.
//
// Exported:  lpipm_solve_f32  (T = float, the ABI entry for InteriorPoint<f32>)
//            lpipm_k_generic_solve_f64  (T = double: test hook -- the SAME kernels against the fp64 oracle, where agreement to
//            1e-8 shows that the generic kernels restate the algorithm; f32 itself is only checkable against an f32 oracle,
//            to f32 accuracy)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "lpipm_internal.hpp"
#include "vec_kernels.hpp"      // ST_* status values

namespace lpipm {
namespace generic {

constexpr int GB = 64;          // block edge of the factorisation and of the GEMM tiles
constexpr int VT = 1024;        // threads of the single-workgroup vector kernels

// ------------------------------------------------------------------------------------------------ GEMM
// C(ti, tj) = beta C + alpha sum_k P[i][k] s[k] Q[j][k]   (row-major operands, K contiguous; s nullable)
// lower != 0: only the tiles with tj <= ti of the M x M result (tile index -> lower triangle, row-major)
template <typename T>
__global__ __launch_bounds__(256) void kg_gemm_nt(const T* __restrict__ P, long long ldp, const T* __restrict__ Q, long long ldq,
                                                  const T* __restrict__ s, T* __restrict__ C, long long ldc, int M, int N, int K,
                                                  T alpha, T beta, int lower, int ntj) {
    __shared__ T Ps[GB][17], Qs[GB][17];
    int ti, tj;
    if (lower) {
        const int t = blockIdx.x;
        int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        ti = i; tj = t - i * (i + 1) / 2;
    } else { ti = blockIdx.x / ntj; tj = blockIdx.x - ti * ntj; }
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = T(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int e = tid; e < GB * 16; e += 256) {
            const int r = e >> 4, kk = e & 15, k = k0 + kk;
            const int rp = ti * GB + r, rq = tj * GB + r;
            Ps[r][kk] = (rp < M && k < K) ? P[(long long)rp * ldp + k] : T(0);
            T q = (rq < N && k < K) ? Q[(long long)rq * ldq + k] : T(0);
            if (s && k < K) q *= s[k];
            Qs[r][kk] = q;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = Ps[ty * 4 + i][kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Qs[tx * 4 + j][kk];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = ti * GB + ty * 4 + i, col = tj * GB + tx * 4 + j;
            if (row < M && col < N) {
                T* c = C + (long long)row * ldc + col;
                *c = beta == T(0) ? alpha * acc[i][j] : beta * *c + alpha * acc[i][j];
            }
        }
}

// ------------------------------------------------------------------------------------------------ Cholesky, diagonal block
// In-place lower Cholesky of the bs x bs block at A (ld) and inv = L^-1 (row-major, stride GB, zeros above the diagonal).
// info: 0, or 1 + global index of the first non-positive pivot (the factorisation goes on, on NaNs).
template <typename T>
__global__ __launch_bounds__(256) void kg_potrf_diag(T* __restrict__ A, long long ld, int bs, T* __restrict__ inv, int* info, int row0) {
    __shared__ T a[GB][GB + 1];
    __shared__ T x[GB][GB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < GB * GB; e += 256) {
        const int r = e / GB, c = e % GB;
        a[r][c] = (r < bs && c <= r) ? A[(long long)r * ld + c] : T(0);
        x[r][c] = T(0);
    }
    __syncthreads();
    for (int j = 0; j < bs; ++j) {
        if (tid == 0) {
            const T d = a[j][j];
            if (!(d > T(0))) atomicCAS(info, 0, row0 + j + 1);
            a[j][j] = sqrt(d);
        }
        __syncthreads();
        const T djj = a[j][j];
        for (int i = j + 1 + tid; i < bs; i += 256) a[i][j] = a[i][j] / djj;
        __syncthreads();
        for (int i = j + 1 + (tid >> 6); i < bs; i += 4)
            for (int k = j + 1 + (tid & 63); k <= i; k += 64) a[i][k] -= a[i][j] * a[k][j];
        __syncthreads();
    }
    // inverse by forward substitution on the identity: thread t owns column t
    if (tid < bs) {
        const int t = tid;
        x[t][t] = T(1) / a[t][t];
        for (int i = t + 1; i < bs; ++i) {
            T sum = T(0);
            for (int k = t; k < i; ++k) sum += a[i][k] * x[k][t];
            x[i][t] = -sum / a[i][i];
        }
    }
    __syncthreads();
    for (int e = tid; e < GB * GB; e += 256) {
        const int r = e / GB, c = e % GB;
        if (r < bs && c <= r) A[(long long)r * ld + c] = a[r][c];
        inv[r * GB + c] = (r < bs && c <= r) ? x[r][c] : T(0);
    }
}

template <typename T>
__global__ void kg_copy2d(T* __restrict__ dst, long long ldd, const T* __restrict__ src, long long lds_, int rows, int cols) {
    const int r = blockIdx.x, c0 = threadIdx.x;
    if (r >= rows) return;
    for (int c = c0; c < cols; c += blockDim.x) dst[(long long)r * ldd + c] = src[(long long)r * lds_ + c];
}

// ------------------------------------------------------------------------------------------------ block substitution
// y = inv . r  (trans == 0) or inv^T . r (trans == 1) on one bs-block, in place
template <typename T>
__global__ __launch_bounds__(GB) void kg_block_mv(const T* __restrict__ inv, int bs, T* __restrict__ r, int trans) {
    __shared__ T v[GB];
    const int t = threadIdx.x;
    v[t] = t < bs ? r[t] : T(0);
    __syncthreads();
    if (t >= bs) return;
    T sum = T(0);
    if (!trans) for (int k = 0; k <= t; ++k) sum += inv[t * GB + k] * v[k];
    else        for (int k = t; k < bs; ++k) sum += inv[k * GB + t] * v[k];
    r[t] = sum;
}
// forward: r[i] -= sum_k L[i][k] y[k]  for the rows below a block (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void kg_sub_below(const T* __restrict__ L, long long ld, int rows, int bs, const T* __restrict__ y,
                                                    T* __restrict__ r) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    T sum = lane < bs ? L[(long long)row * ld + lane] * y[lane] : T(0);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) r[row] -= sum;
}
// backward: r[i] -= sum_k L[k][i] y[k]  for the columns left of a block (one thread per column i)
template <typename T>
__global__ __launch_bounds__(256) void kg_sub_left(const T* __restrict__ Lrow, long long ld, int cols, int bs, const T* __restrict__ y,
                                                   T* __restrict__ r) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cols) return;
    T sum = T(0);
    for (int k = 0; k < bs; ++k) sum += Lrow[(long long)k * ld + i] * y[k];
    r[i] -= sum;
}

// ------------------------------------------------------------------------------------------------ GEMV
// y[i] = sum_k A[i][k] w[k]  (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void kg_gemv_n(const T* __restrict__ A, long long lda, int m, int n, const T* __restrict__ w,
                                                 T* __restrict__ y) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    T sum = T(0);
    for (int k = lane; k < n; k += 64) sum += A[(long long)row * lda + k] * w[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) y[row] = sum;
}
// partial[s][k] = sum over row chunk s (128 rows) of A[i][k] v[i];  kg_fold adds the chunks in order
template <typename T>
__global__ __launch_bounds__(256) void kg_gemv_t(const T* __restrict__ A, long long lda, int m, int n, const T* __restrict__ v,
                                                 T* __restrict__ partial) {
    const int k = blockIdx.x * 256 + threadIdx.x, r0 = blockIdx.y * 128;
    if (k >= n) return;
    const int r1 = r0 + 128 < m ? r0 + 128 : m;
    T sum = T(0);
    for (int i = r0; i < r1; ++i) sum += A[(long long)i * lda + k] * v[i];
    partial[(long long)blockIdx.y * n + k] = sum;
}
template <typename T>
__global__ __launch_bounds__(256) void kg_fold(const T* __restrict__ partial, int nsplit, int n, T* __restrict__ u) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    T sum = T(0);
    for (int s = 0; s < nsplit; ++s) sum += partial[(long long)s * n + k];
    u[k] = sum;
}

// ------------------------------------------------------------------------------------------------ vector stage
template <typename T> struct GStatus { T alpha, rho_p, rho_d, rho_A, rho_g, rho_mu, obj; int status, nan_pq, potrf_info, pad; };
template <typename T> struct GScal {    // device scalars of one solve
    T tau, kappa, mu, rG, gamma, eta, rhat_g, rhat_tk, dtau, dkappa, alpha_pred, alpha, cp, bq, rp0, rd0, rg0, rmu0, c0;
};
template <typename T> struct GVec {
    int m, n;
    const T *A, *b, *c;
    T *x, *y, *z, *dinv, *xs, *r1, *rP, *rD, *p, *q, *u, *v, *dx, *dy, *dz, *dxdz, *w, *rr, *Ax, *ATy, *t;
    GScal<T>* S;
    GStatus<T>* st;
    int* info;
};

template <typename T> __device__ T block_sum(T v, T* red) {          // fixed-order tree over the VT threads
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = VT / 2; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const T r = red[0];
    __syncthreads();
    return r;
}
template <typename T> __device__ T block_min(T v, T* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = VT / 2; s >= 1; s >>= 1) { if (tid < s) red[tid] = fmin(red[tid], red[tid + s]); __syncthreads(); }
    const T r = red[0];
    __syncthreads();
    return r;
}

// FeasiblePoint::blind_start, feasible_point.rs:24-31
template <typename T> __global__ __launch_bounds__(VT) void kv_init(GVec<T> a, T c0) {
    for (int j = threadIdx.x; j < a.n; j += VT) { a.x[j] = T(1); a.z[j] = T(1); }
    for (int i = threadIdx.x; i < a.m; i += VT) a.y[i] = T(0);
    if (threadIdx.x == 0) { a.S->tau = T(1); a.S->kappa = T(1); a.S->c0 = c0; a.st->nan_pq = 0; *a.info = 0; }
}

// Residuals::calculate (residual.rs:13-44), Indicators (indicators.rs:37-83) at the current point (Ax, ATy given), and the
// scalars the next get_delta starts from (feasible_point.rs:119-125): r_P, r_D are kept for it.
template <typename T> __global__ __launch_bounds__(VT) void kv_resid(GVec<T> a, int is_init, int ip_next, T tol) {
    __shared__ T red[VT];
    GScal<T>& S = *a.S;
    const T tau = S.tau, kappa = S.kappa;
    T sp = 0, by = 0, sd = 0, cx = 0, xz = 0, cxt = 0;
    for (int i = threadIdx.x; i < a.m; i += VT) {
        const T r = a.b[i] * tau - a.Ax[i];                           // residual.rs:22-24, feasible_point.rs:122
        a.rP[i] = r;
        sp += r * r;
        by += a.b[i] * a.y[i];
    }
    for (int j = threadIdx.x; j < a.n; j += VT) {
        const T xj = a.x[j], zj = a.z[j], cj = a.c[j];
        const T r = cj * tau - a.ATy[j] - zj;                        // residual.rs:25-26, feasible_point.rs:123
        a.rD[j] = r;
        sd += r * r;
        cx += cj * xj;
        xz += xj * zj;
        cxt += cj * (xj / tau);                                       // indicators.rs:41
    }
    sp = block_sum(sp, red); by = block_sum(by, red); sd = block_sum(sd, red);
    cx = block_sum(cx, red); xz = block_sum(xz, red); cxt = block_sum(cxt, red);
    if (threadIdx.x != 0) return;
    const T rho_p = sqrt(sp), rho_d = sqrt(sd);                      // residual.rs:34-35
    const T rho_g = fabs(kappa + cx - by);                            // :27-29,36
    const T rho_mu = (xz + tau * kappa) / (T)(a.n + 1);               // :30-32,37
    if (is_init) { S.rp0 = rho_p; S.rd0 = rho_d; S.rg0 = rho_g; S.rmu0 = rho_mu; }    // feasible_point.rs:32
    GStatus<T>& st = *a.st;
    const T ip_ = rho_p / fmax(S.rp0, T(1)), id_ = rho_d / fmax(S.rd0, T(1));         // indicators.rs:47-48
    const T ig_ = rho_g / fmax(S.rg0, T(1)), imu = rho_mu / S.rmu0;                   // :50-51
    const T rho_A = fabs(cx - by) / (tau + fabs(by));                                 // :43-44
    st.alpha = is_init ? T(1) : S.alpha;
    st.rho_p = ip_; st.rho_d = id_; st.rho_A = rho_A; st.rho_g = ig_; st.rho_mu = imu; st.obj = cxt + S.c0;
    int status = ST_UNFINISHED;
    if (!is_init) {                                                                   // indicators.rs:66-83
        const bool tau_small = tau < tol * fmax(kappa, T(1));
        const bool inf1 = (ip_ < tol && id_ < tol && ig_ < tol) && tau_small;
        const bool inf2 = imu < tol && tau_small;
        if (inf1 || inf2) status = by > tol ? ST_INFEASIBLE : ST_UNBOUNDED;
        else if (ip_ < tol && id_ < tol && rho_A < tol) status = ST_OPTIMAL;
    }
    st.status = status;
    st.potrf_info = *a.info;
    const T gamma = ip_next ? T(1) : T(0);                                            // feasible_point.rs:119
    const T eta = ip_next ? T(1) : T(1) - gamma;                                      // :120
    S.rG = cx - by + kappa;                                                           // :124
    S.mu = (xz + tau * kappa) / (T)(a.n + 1);                                         // :125
    S.gamma = gamma; S.eta = eta;
    S.rhat_g = S.rG * eta;                                                            // rhat.rs:31
    S.rhat_tk = gamma * S.mu - tau * kappa;                                           // rhat.rs:33
}

// Dinv (newton_equations.rs:54), Rhat::predictor (rhat.rs:29-32), r1 of the second sym_solve (:188)
template <typename T> __global__ __launch_bounds__(VT) void kv_pred(GVec<T> a) {
    const GScal<T>& S = *a.S;
    const T gm = S.gamma * S.mu, eta = S.eta;
    for (int j = threadIdx.x; j < a.n; j += VT) {
        const T xj = a.x[j], zj = a.z[j];
        a.dinv[j] = xj / zj;
        const T xs = (xj * T(-1)) * zj + gm;
        a.xs[j] = xs;
        a.r1[j] = a.rD[j] * eta - xs / xj;
    }
}
// w = Dinv o r1 (sym_solve prologue, newton_equations.rs:220); which: 0 -> r1 = c, 1 -> r1 = a.r1
template <typename T> __global__ __launch_bounds__(VT) void kv_w(GVec<T> a, int which) {
    for (int j = threadIdx.x; j < a.n; j += VT) a.w[j] = a.dinv[j] * (which ? a.r1[j] : a.c[j]);
}
// rr = r2 + A.w (newton_equations.rs:220): which 0 -> r2 = b; 1 -> r2 = rhat.p = rP * eta (rhat.rs:29 / :69)
template <typename T> __global__ __launch_bounds__(VT) void kv_rhs(GVec<T> a, int which) {
    const T eta = a.S->eta;
    for (int i = threadIdx.x; i < a.m; i += VT) a.rr[i] = (which ? a.rP[i] * eta : a.b[i]) + a.Ax[i];
}
// sym_solve epilogue (newton_equations.rs:223): out = Dinv o (A^T v - r1); v (the solve's result, in rr) is kept in vout
template <typename T> __global__ __launch_bounds__(VT) void kv_epi(GVec<T> a, int which) {
    T* out = which ? a.u : a.p;
    T* vout = which ? a.v : a.q;
    int nan = 0;
    for (int j = threadIdx.x; j < a.n; j += VT) {
        const T val = a.dinv[j] * (a.ATy[j] - (which ? a.r1[j] : a.c[j]));
        out[j] = val;
        if (!which) nan |= (val != val);
    }
    for (int i = threadIdx.x; i < a.m; i += VT) { const T val = a.rr[i]; vout[i] = val; if (!which) nan |= (val != val); }
    if (nan) atomicOr(&a.st->nan_pq, 1);                              // newton_equations.rs:190-194
}
// Delta::compute (delta.rs:21-49), get_step_size (feasible_point.rs:53-72); phase 0: predictor -> alpha_pred, update_gamma
// (:156-165), eta (:136), Rhat::corrector (rhat.rs:37-75) and the corrector's r1; phase 1: the iteration's delta, the
// step length (mod.rs:216-221) and do_step (feasible_point.rs:76-106).
template <typename T> __global__ __launch_bounds__(VT) void kv_delta(GVec<T> a, int phase, int ip, T alpha0) {
    __shared__ T red[VT];
    __shared__ T sh[4];
    GScal<T>& S = *a.S;
    T cu = 0, bv = 0, cp = 0, bq = 0;
    for (int j = threadIdx.x; j < a.n; j += VT) { cu += a.c[j] * a.u[j]; cp += a.c[j] * a.p[j]; }
    for (int i = threadIdx.x; i < a.m; i += VT) { bv += a.b[i] * a.v[i]; bq += a.b[i] * a.q[i]; }
    cu = block_sum(cu, red); bv = block_sum(bv, red); cp = block_sum(cp, red); bq = block_sum(bq, red);
    const T tau = S.tau, kappa = S.kappa;
    const T d_tau = (S.rhat_g + T(1) / tau * S.rhat_tk - (-cu + bv)) / (T(1) / tau * kappa + (-cp + bq));     // delta.rs:29-32
    const T d_kappa = T(1) / tau * (S.rhat_tk - kappa * d_tau);                                                // :38
    T ax = T(1), az = T(1);
    for (int j = threadIdx.x; j < a.n; j += VT) {
        const T xj = a.x[j], zj = a.z[j];
        const T dx = a.u[j] + a.p[j] * d_tau;                                                                  // :33
        const T dz = (a.xs[j] - zj * dx) / xj;                                                                 // :37
        if (dx < T(0)) ax = fmin(ax, xj / -dx);
        if (dz < T(0)) az = fmin(az, zj / -dz);
        a.dx[j] = dx; a.dz[j] = dz;
    }
    for (int i = threadIdx.x; i < a.m; i += VT) a.dy[i] = a.v[i] + a.q[i] * d_tau;                             // :34
    ax = block_min(ax, red); az = block_min(az, red);
    const T at = d_tau < T(0) ? fmin(T(1), tau / -d_tau) : T(1);
    const T ak = d_kappa < T(0) ? fmin(T(1), kappa / -d_kappa) : T(1);
    const T amin = fmin(fmin(fmin(fmin(T(1), ax), at), az), ak);                                              // feasible_point.rs:66-71
    if (phase == 0) {
        const T alpha = amin * T(1);                                                                           // :134
        const T mu = S.mu;
        const T gamma = ip ? T(10) : (T(1) - alpha) * (T(1) - alpha) * fmin(T(0.1), T(1) - alpha);             // :156-165
        const T eta = ip ? T(1) : T(1) - gamma;                                                                // :136
        const T alpha_2 = alpha * alpha;
        T tk;
        if (ip) tk = (T(1) - alpha) * gamma * mu - tau * kappa - alpha_2 * d_tau * d_kappa;                    // rhat.rs:57-59
        else    tk = gamma * mu - tau * kappa - d_tau * d_kappa;                                               // rhat.rs:65
        for (int j = threadIdx.x; j < a.n; j += VT) {
            const T xj = a.x[j], zj = a.z[j], pr = a.dx[j] * a.dz[j];
            T xs;
            if (ip) xs = (xj * T(-1)) * zj - pr * alpha_2 + (T(1) - alpha) * gamma * mu;                       // rhat.rs:52-56
            else    xs = (xj * T(-1)) * zj + gamma * mu - pr;                                                  // rhat.rs:62-64
            a.xs[j] = xs;
            a.r1[j] = a.rD[j] * eta - xs / xj;                                                                 // newton_equations.rs:188
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            S.alpha_pred = alpha; S.gamma = gamma; S.eta = eta;
            S.rhat_g = S.rG * eta;                                                                             // rhat.rs:71
            S.rhat_tk = tk;
        }
    } else {
        const T alpha = ip ? T(1) : amin * alpha0;                                                             // mod.rs:216-221
        for (int j = threadIdx.x; j < a.n; j += VT) {                                                          // do_step
            T xn = a.x[j] + a.dx[j] * alpha, zn = a.z[j] + a.dz[j] * alpha;
            if (ip) { xn = fmax(xn, T(1)); zn = fmax(zn, T(1)); }
            a.x[j] = xn; a.z[j] = zn;
        }
        for (int i = threadIdx.x; i < a.m; i += VT) a.y[i] = a.y[i] + a.dy[i] * alpha;
        __syncthreads();
        if (threadIdx.x == 0) {
            T tn = tau + d_tau * alpha, kn = kappa + d_kappa * alpha;
            if (ip) { tn = fmax(tn, T(1)); kn = fmax(kn, T(1)); }
            S.tau = tn; S.kappa = kn; S.alpha = alpha;
        }
    }
    (void)sh;
}
// x / tau (mod.rs:231,238) and fun = c.x + c0 (mod.rs:165, linear_program.rs:61-63)
template <typename T> __global__ __launch_bounds__(VT) void kv_final(GVec<T> a, T* xout, T* fun) {
    __shared__ T red[VT];
    const T tau = a.S->tau;
    T s = 0;
    for (int j = threadIdx.x; j < a.n; j += VT) { const T v = a.x[j] / tau; xout[j] = v; s += a.c[j] * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) *fun = s + a.S->c0;
}

// ------------------------------------------------------------------------------------------------ host side
template <typename T> struct Work {
    std::vector<void*> allocs;
    hipStream_t st = nullptr;
    ~Work() { for (void* p : allocs) (void)hipFree(p); }
    template <typename U> int take(U** out, size_t count) {
        void* p = nullptr;
        LP_HIP(hipMalloc(&p, (count ? count : 1) * sizeof(U)));
        allocs.push_back(p);
        LP_HIP(hipMemsetAsync(p, 0, (count ? count : 1) * sizeof(U), st));
        *out = (U*)p;
        return LPIPM_OK;
    }
};
#define G_TRY(expr) do { int rc__ = (expr); if (rc__ != LPIPM_OK) return rc__; } while (0)

template <typename T>
static int gemm_nt(hipStream_t st, const T* P, long long ldp, const T* Q, long long ldq, const T* s, T* C, long long ldc, int M, int N,
                   int K, T alpha, T beta, bool lower) {
    const int tm = (M + GB - 1) / GB, tn = (N + GB - 1) / GB;
    const int ntiles = lower ? tm * (tm + 1) / 2 : tm * tn;
    if (ntiles <= 0) return LPIPM_OK;
    hipLaunchKernelGGL(kg_gemm_nt<T>, dim3(ntiles), dim3(256), 0, st, P, ldp, Q, ldq, s, C, ldc, M, N, K, alpha, beta, lower ? 1 : 0, tn);
    LP_HIP(hipGetLastError());
    return LPIPM_OK;
}
// newton_equations.rs:129-131: M = L.L^T in place (lower), inv = the inverses of the diagonal 64-blocks
template <typename T>
static int potrf(hipStream_t st, T* M, int m, T* inv, T* ltmp, int* info) {
    const int nb = (m + GB - 1) / GB;
    for (int jb = 0; jb < nb; ++jb) {
        const int o = jb * GB, bs = m - o < GB ? m - o : GB, rem = m - o - bs;
        T* d = M + (long long)o * m + o;
        hipLaunchKernelGGL(kg_potrf_diag<T>, dim3(1), dim3(256), 0, st, d, (long long)m, bs, inv + (size_t)jb * GB * GB, info, o);
        LP_HIP(hipGetLastError());
        if (rem <= 0) break;
        T* a21 = M + (long long)(o + bs) * m + o;
        G_TRY(gemm_nt<T>(st, a21, m, inv + (size_t)jb * GB * GB, GB, nullptr, ltmp, GB, rem, bs, bs, T(1), T(0), false));   // L21 = A21 . inv^T
        hipLaunchKernelGGL(kg_copy2d<T>, dim3(rem), dim3(64), 0, st, a21, (long long)m, ltmp, (long long)GB, rem, bs);
        LP_HIP(hipGetLastError());
        T* a22 = M + (long long)(o + bs) * m + (o + bs);
        G_TRY(gemm_nt<T>(st, a21, m, a21, m, nullptr, a22, m, rem, rem, bs, T(-1), T(1), true));                              // A22 -= L21 . L21^T
    }
    return LPIPM_OK;
}
// newton_equations.rs:151-169: r <- L^-T (L^-1 r)
template <typename T>
static int chol_solve(hipStream_t st, const T* L, int m, const T* inv, T* r) {
    const int nb = (m + GB - 1) / GB;
    for (int jb = 0; jb < nb; ++jb) {
        const int o = jb * GB, bs = m - o < GB ? m - o : GB, rem = m - o - bs;
        hipLaunchKernelGGL(kg_block_mv<T>, dim3(1), dim3(GB), 0, st, inv + (size_t)jb * GB * GB, bs, r + o, 0);
        if (rem > 0)
            hipLaunchKernelGGL(kg_sub_below<T>, dim3((rem + 3) / 4), dim3(256), 0, st, L + (long long)(o + bs) * m + o, (long long)m, rem, bs,
                               r + o, r + o + bs);
    }
    for (int jb = nb - 1; jb >= 0; --jb) {
        const int o = jb * GB, bs = m - o < GB ? m - o : GB;
        hipLaunchKernelGGL(kg_block_mv<T>, dim3(1), dim3(GB), 0, st, inv + (size_t)jb * GB * GB, bs, r + o, 1);
        if (o > 0)
            hipLaunchKernelGGL(kg_sub_left<T>, dim3((o + 255) / 256), dim3(256), 0, st, L + (long long)o * m, (long long)m, o, bs, r + o, r);
    }
    LP_HIP(hipGetLastError());
    return LPIPM_OK;
}

template <typename T>
static int generic_solve(lpipm_ctx_device dev, uint64_t m64, uint64_t n64, const T* A, uint64_t lda, const T* b, const T* c, T c0,
                         const lpipm_opts* o, T* x_out, T* fun_out, uint64_t* its_out, GStatus<T>* log) {
    if (!o || !A || !b || !c || !x_out) return LPIPM_ERR_BAD_ARGUMENT;
    if (!(o->alpha0 > 0.0) || !(o->alpha0 < 1.0) || !(o->tol > 0.0)) return LPIPM_INVALID_PARAMETER;     // mod.rs:118-128
    if (o->solver_type != LPIPM_SOLVER_CHOLESKY) return LPIPM_ERR_UNSUPPORTED;                           // the QR arms are fp64 only
    if (m64 == 0) return LPIPM_UNCONSTRAINED;
    if (n64 == 0 || lda < n64 || m64 > 16384 || n64 > (1u << 22)) return LPIPM_ERR_BAD_ARGUMENT;
    const int m = (int)m64, n = (int)n64;
    LP_HIP(hipSetDevice(dev.device));
    Work<T> w;
    w.st = dev.stream;
    hipStream_t st = w.st;
    GVec<T> a{};
    a.m = m; a.n = n;
    T *dA, *db, *dc, *M, *inv, *ltmp, *part, *xout, *dfun;
    const int nb = (m + GB - 1) / GB, nsplit = (m + 127) / 128;
    G_TRY(w.take(&dA, (size_t)m * n)); G_TRY(w.take(&db, m)); G_TRY(w.take(&dc, n));
    G_TRY(w.take(&a.x, n)); G_TRY(w.take(&a.y, m)); G_TRY(w.take(&a.z, n)); G_TRY(w.take(&a.dinv, n)); G_TRY(w.take(&a.xs, n));
    G_TRY(w.take(&a.r1, n)); G_TRY(w.take(&a.rP, m)); G_TRY(w.take(&a.rD, n)); G_TRY(w.take(&a.p, n)); G_TRY(w.take(&a.q, m));
    G_TRY(w.take(&a.u, n)); G_TRY(w.take(&a.v, m)); G_TRY(w.take(&a.dx, n)); G_TRY(w.take(&a.dy, m)); G_TRY(w.take(&a.dz, n));
    G_TRY(w.take(&a.w, n)); G_TRY(w.take(&a.rr, m)); G_TRY(w.take(&a.Ax, m)); G_TRY(w.take(&a.ATy, n));
    G_TRY(w.take(&a.S, 1)); G_TRY(w.take(&a.st, 1)); G_TRY(w.take(&a.info, 1));
    G_TRY(w.take(&M, (size_t)m * m)); G_TRY(w.take(&inv, (size_t)nb * GB * GB)); G_TRY(w.take(&ltmp, (size_t)m * GB));
    G_TRY(w.take(&part, (size_t)nsplit * n)); G_TRY(w.take(&xout, n)); G_TRY(w.take(&dfun, 1));
    a.A = dA; a.b = db; a.c = dc;
    LP_HIP(hipMemcpy2DAsync(dA, (size_t)n * sizeof(T), A, (size_t)lda * sizeof(T), (size_t)n * sizeof(T), (size_t)m, hipMemcpyHostToDevice, st));
    LP_HIP(hipMemcpyAsync(db, b, (size_t)m * sizeof(T), hipMemcpyHostToDevice, st));
    LP_HIP(hipMemcpyAsync(dc, c, (size_t)n * sizeof(T), hipMemcpyHostToDevice, st));
    auto gemv_n = [&](const T* vec) -> int {        // a.Ax = A . vec
        hipLaunchKernelGGL(kg_gemv_n<T>, dim3((m + 3) / 4), dim3(256), 0, st, dA, (long long)n, m, n, vec, a.Ax);
        LP_HIP(hipGetLastError());
        return LPIPM_OK;
    };
    auto gemv_t = [&](const T* vec) -> int {        // a.ATy = A^T . vec
        hipLaunchKernelGGL(kg_gemv_t<T>, dim3((n + 255) / 256, nsplit), dim3(256), 0, st, dA, (long long)n, m, n, vec, part);
        hipLaunchKernelGGL(kg_fold<T>, dim3((n + 255) / 256), dim3(256), 0, st, part, nsplit, n, a.ATy);
        LP_HIP(hipGetLastError());
        return LPIPM_OK;
    };
    auto resid = [&](int is_init, int ip_next) -> int {
        G_TRY(gemv_n(a.x));                                                        // residual.rs:23
        G_TRY(gemv_t(a.y));                                                        // residual.rs:25
        hipLaunchKernelGGL(kv_resid<T>, dim3(1), dim3(VT), 0, st, a, is_init, ip_next, (T)o->tol);
        LP_HIP(hipGetLastError());
        return LPIPM_OK;
    };
    // sym_solve (newton_equations.rs:214-225): which 0 -> (p, q) = sym_solve(c, b); 1 -> (u, v) = sym_solve(r1, rhat.p)
    auto sym_solve = [&](int which) -> int {
        hipLaunchKernelGGL(kv_w<T>, dim3(1), dim3(VT), 0, st, a, which);
        G_TRY(gemv_n(a.w));
        hipLaunchKernelGGL(kv_rhs<T>, dim3(1), dim3(VT), 0, st, a, which);
        G_TRY(chol_solve<T>(st, M, m, inv, a.rr));                                 // :221
        G_TRY(gemv_t(a.rr));
        hipLaunchKernelGGL(kv_epi<T>, dim3(1), dim3(VT), 0, st, a, which);         // :223
        LP_HIP(hipGetLastError());
        return LPIPM_OK;
    };
    GStatus<T> hs{};
    auto read_status = [&]() -> int {
        LP_HIP(hipMemcpyAsync(&hs, a.st, sizeof(hs), hipMemcpyDeviceToHost, st));
        LP_HIP(hipStreamSynchronize(st));
        return LPIPM_OK;
    };
    hipLaunchKernelGGL(kv_init<T>, dim3(1), dim3(VT), 0, st, a, c0);               // feasible_point.rs:24-31
    G_TRY(resid(1, o->ip ? 1 : 0));                                                // feasible_point.rs:32, mod.rs:206
    G_TRY(read_status());
    if (o->disp) {                                                                 // mod.rs:208-211
        printf("alpha     \trho_p     \trho_d     \trho_g     \trho_mu    \tobj       \n");
        printf("1.00000000\t%.8f\t%.8f\t%.8f\t%.8f\t%8.3f\n", (double)hs.rho_p, (double)hs.rho_d, (double)hs.rho_g, (double)hs.rho_mu, (double)hs.obj);
    }
    int ip = o->ip ? 1 : 0, ret = LPIPM_ITERATION_LIMIT;
    uint64_t iteration = 0;
    for (iteration = 1; iteration <= o->max_iter; ++iteration) {                   // mod.rs:213
        // get_delta, feasible_point.rs:110-152
        hipLaunchKernelGGL(kv_pred<T>, dim3(1), dim3(VT), 0, st, a);
        G_TRY(gemm_nt<T>(st, dA, n, dA, n, a.dinv, M, m, m, m, n, T(1), T(0), true));   // newton_equations.rs:55-57
        LP_HIP(hipMemsetAsync(a.info, 0, sizeof(int), st));
        G_TRY(potrf<T>(st, M, m, inv, ltmp, a.info));                              // :129-131
        G_TRY(sym_solve(0));                                                       // :187
        G_TRY(sym_solve(1));                                                       // :188
        hipLaunchKernelGGL(kv_delta<T>, dim3(1), dim3(VT), 0, st, a, 0, ip, (T)o->alpha0);   // predictor -> corrector set-up
        G_TRY(sym_solve(1));                                                       // feasible_point.rs:149 (p, q unchanged: same inputs, same factor)
        hipLaunchKernelGGL(kv_delta<T>, dim3(1), dim3(VT), 0, st, a, 1, ip, (T)o->alpha0);   // delta, step length, do_step
        G_TRY(resid(0, 0));                                                        // mod.rs:225
        G_TRY(read_status());
        if (hs.potrf_info != 0 || hs.nan_pq) { ret = LPIPM_NUMERICAL_PROBLEM; break; }   // newton_equations.rs:58-63, :190-194
        ip = 0;                                                                    // mod.rs:223
        if (o->disp)
            printf("%.8f\t%.8f\t%.8f\t%.8f\t%.8f\t%8.3f\n", (double)hs.alpha, (double)hs.rho_p, (double)hs.rho_d, (double)hs.rho_g, (double)hs.rho_mu, (double)hs.obj);
        if (log) log[iteration - 1] = hs;
        if (hs.status == ST_OPTIMAL) { ret = LPIPM_OK; break; }                    // mod.rs:231-233
        if (hs.status == ST_INFEASIBLE) { ret = LPIPM_INFEASIBLE; break; }
        if (hs.status == ST_UNBOUNDED) { ret = LPIPM_UNBOUNDED; break; }
    }
    if (ret == LPIPM_ITERATION_LIMIT) iteration = o->max_iter;
    if (ret == LPIPM_OK || ret == LPIPM_ITERATION_LIMIT) {
        hipLaunchKernelGGL(kv_final<T>, dim3(1), dim3(VT), 0, st, a, xout, dfun);
        LP_HIP(hipMemcpyAsync(x_out, xout, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, st));
        T fun = T(0);
        LP_HIP(hipMemcpyAsync(&fun, dfun, sizeof(T), hipMemcpyDeviceToHost, st));
        LP_HIP(hipStreamSynchronize(st));
        if (fun_out) *fun_out = fun;
    } else {
        LP_HIP(hipStreamSynchronize(st));
    }
    if (its_out) *its_out = iteration;
    return ret;
}

}  // namespace generic
}  // namespace lpipm

using namespace lpipm;

// interior_point/mod.rs:161-168 for F = f32 (src/float.rs:42-43).  Host arrays in, host arrays out; the problem is uploaded,
// solved and released inside the call (the f32 instantiation keeps no state in the context).  log (nullable): max_iter rows.
extern "C" int lpipm_solve_f32(lpipm_ctx* ctx, uint64_t m, uint64_t n, const float* A, uint64_t lda, const float* b, const float* c,
                               float c0, const lpipm_opts* opts, float* x_slack_out, float* fun_out, uint64_t* iterations_out,
                               lpipm_iter_row_f32* log) {
    if (!ctx) return LPIPM_ERR_BAD_ARGUMENT;
    std::vector<generic::GStatus<float>> rows(log && opts ? (size_t)opts->max_iter : 0);
    uint64_t its = 0;
    const int rc = generic::generic_solve<float>(lpipm_ctx_device_of(ctx), m, n, A, lda, b, c, c0, opts, x_slack_out, fun_out, &its,
                                                 rows.empty() ? nullptr : rows.data());
    if (iterations_out) *iterations_out = its;
    for (uint64_t i = 0; log && i < its && i < rows.size() && rc != LPIPM_NUMERICAL_PROBLEM; ++i) {
        const auto& r = rows[i];
        log[i].alpha = r.alpha; log[i].rho_p = r.rho_p; log[i].rho_d = r.rho_d; log[i].rho_A = r.rho_A;
        log[i].rho_g = r.rho_g; log[i].rho_mu = r.rho_mu; log[i].obj = r.obj;
    }
    return rc;
}

// Test hook: the same generic kernels with T = double (see the head of this file).
extern "C" int lpipm_k_generic_solve_f64(lpipm_ctx* ctx, uint64_t m, uint64_t n, const double* A, uint64_t lda, const double* b,
                                         const double* c, double c0, const lpipm_opts* opts, double* x_slack_out, double* fun_out,
                                         uint64_t* iterations_out, lpipm_iter_row* log) {
    if (!ctx) return LPIPM_ERR_BAD_ARGUMENT;
    std::vector<generic::GStatus<double>> rows(log && opts ? (size_t)opts->max_iter : 0);
    uint64_t its = 0;
    const int rc = generic::generic_solve<double>(lpipm_ctx_device_of(ctx), m, n, A, lda, b, c, c0, opts, x_slack_out, fun_out, &its,
                                                  rows.empty() ? nullptr : rows.data());
    if (iterations_out) *iterations_out = its;
    for (uint64_t i = 0; log && i < its && i < rows.size() && rc != LPIPM_NUMERICAL_PROBLEM; ++i) {
        const auto& r = rows[i];
        log[i].alpha = r.alpha; log[i].rho_p = r.rho_p; log[i].rho_d = r.rho_d; log[i].rho_A = r.rho_A;
        log[i].rho_g = r.rho_g; log[i].rho_mu = r.rho_mu; log[i].obj = r.obj;
    }
    return rc;
}
