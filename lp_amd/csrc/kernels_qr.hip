// kernels_qr.hip -- EquationSolverType::{Inverse, LeastSquares}: solve M v = r through a Householder
// QR factorisation of M on device (newton_equations.rs:133-149 `M.qr()`, :155-166 `solve_into`).
//
// In the reference's default backend both arms are the same computation for the square M of the
// normal equations: `least_squares` takes the `nrows >= ncols` branch (:138-139) and `solve_into` is
// R^-1 Q^T b.  These are the slow, robust fall-back arms (the reference says "about twice as
// expensive", :43-45); here they are a straightforward column-by-column Householder QR:
//   qr_house_kernel  (1 workgroup)  : reflector k from column k  -> v below the diagonal, R[k][k], tau[k]
//   qr_apply_kernel  (column blocks): trailing columns  a_j -= tau (v^T a_j) v        (2 passes over rows)
//   qr_solve_kernel  (1 workgroup)  : y = Q^T r (reflectors in order), then back-substitution R x = y
// Correct and HBM-streaming but unblocked (no compact-WY / MFMA): ~0.2 s per factorisation at
// m = 4096 against 2.5 ms for the Cholesky arm.  A zero column norm or a zero R[k][k] sets *info
// (-> NumericalProblem, where the reference's `?` on qr()/solve_into() lands, :134,:155-166).
#include "lpipm_internal.hpp"

namespace lpipm {

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double block_sum_1024(double v, double* sm) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += sm[w];   // fixed order: reproducible
    return s;
}

// M[j][i] = M[i][j] for j < i: A.D.A^T only forms the lower triangle, QR needs the whole matrix.
__global__ __launch_bounds__(256) void symmetrize_kernel(double* __restrict__ M, long long ld, int mp) {
    __shared__ double t[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) t[r][tx] = M[(long long)(bi * 32 + r) * ld + bj * 32 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int row = bj * 32 + r, col = bi * 32 + tx;   // transposed position
        if (col > row) M[(long long)row * ld + col] = t[tx][r];
    }
}

// Householder reflector of column k (rows k..mp-1).  H = I - tau v v^T with v[k] = 1.
__global__ __launch_bounds__(1024) void qr_house_kernel(double* __restrict__ M, long long ld, int mp, int k,
                                                        double* __restrict__ tau, int32_t* info) {
    __shared__ double sm[16];
    double s = 0.0;
    for (int i = k + 1 + threadIdx.x; i < mp; i += blockDim.x) {
        const double a = M[(long long)i * ld + k];
        s += a * a;
    }
    const double sigma = block_sum_1024(s, sm);
    const double akk = M[(long long)k * ld + k];
    const double nrm = sqrt(akk * akk + sigma);
    if (!(nrm > 0.0)) {   // zero (or NaN) column: singular
        if (threadIdx.x == 0) { atomicCAS((int*)info, 0, k + 1); tau[k] = 0.0; }
        return;
    }
    const double alpha = akk > 0.0 ? -nrm : nrm;
    const double v0 = akk - alpha;
    // tau for the v0 = 1 scaling: 2 v0^2 / (v0^2 + sigma)
    const double t = 2.0 * v0 * v0 / (v0 * v0 + sigma);
    __syncthreads();
    for (int i = k + 1 + threadIdx.x; i < mp; i += blockDim.x) M[(long long)i * ld + k] /= v0;
    if (threadIdx.x == 0) { M[(long long)k * ld + k] = alpha; tau[k] = t; }
}

// Trailing columns j in (k, mp): 64 columns per workgroup, 4 row groups reduce through LDS.
__global__ __launch_bounds__(256) void qr_apply_kernel(double* __restrict__ M, long long ld, int mp, int k,
                                                       const double* __restrict__ tau) {
    __shared__ double part[4][64];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = k + 1 + blockIdx.x * 64 + cx;
    const bool act = j < mp;
    const double t = tau[k];
    double s = 0.0;
    if (act) {
        if (rg == 0) s = M[(long long)k * ld + j];
        for (int i = k + 1 + rg; i < mp; i += 4) s += M[(long long)i * ld + k] * M[(long long)i * ld + j];
    }
    part[rg][cx] = s;
    __syncthreads();
    const double w = t * ((part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]));
    if (!act) return;
    if (rg == 0) M[(long long)k * ld + j] -= w;
    for (int i = k + 1 + rg; i < mp; i += 4) M[(long long)i * ld + j] -= w * M[(long long)i * ld + k];
}

// x = R^-1 Q^T r for one right-hand side held in LDS (mp <= 16384).  rhs/out: global vectors of mp.
__global__ __launch_bounds__(1024) void qr_solve_kernel(const double* __restrict__ M, long long ld, int mp,
                                                        const double* __restrict__ tau, double* __restrict__ x,
                                                        int32_t* info) {
    extern __shared__ __attribute__((aligned(16))) double y[];
    __shared__ double sm[16];
    for (int i = threadIdx.x; i < mp; i += blockDim.x) y[i] = x[i];
    __syncthreads();
    for (int k = 0; k < mp; ++k) {   // y <- H_k y
        double s = 0.0;
        for (int i = k + 1 + threadIdx.x; i < mp; i += blockDim.x) s += M[(long long)i * ld + k] * y[i];
        const double w = tau[k] * (block_sum_1024(s, sm) + y[k]);
        __syncthreads();
        for (int i = k + 1 + threadIdx.x; i < mp; i += blockDim.x) y[i] -= w * M[(long long)i * ld + k];
        if (threadIdx.x == 0) y[k] -= w;
        __syncthreads();
    }
    for (int i = mp - 1; i >= 0; --i) {   // R x = y, rows of R are contiguous
        double s = 0.0;
        for (int j = i + 1 + threadIdx.x; j < mp; j += blockDim.x) s += M[(long long)i * ld + j] * y[j];
        const double tot = block_sum_1024(s, sm);
        if (threadIdx.x == 0) {
            const double r = M[(long long)i * ld + i];
            if (r == 0.0) atomicCAS((int*)info, 0, i + 1);
            y[i] = (y[i] - tot) / r;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < mp; i += blockDim.x) x[i] = y[i];
}

hipError_t launch_qr_factor(double* M, int64_t ld, int mp, double* tau, int32_t* info, hipStream_t st) {
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(symmetrize_kernel, dim3(mp / 32, mp / 32), dim3(256), 0, st, M, (long long)ld, mp);
    for (int k = 0; k < mp; ++k) {
        hipLaunchKernelGGL(qr_house_kernel, dim3(1), dim3(1024), 0, st, M, (long long)ld, mp, k, tau, info);
        const int ncols = mp - k - 1;
        if (ncols > 0)
            hipLaunchKernelGGL(qr_apply_kernel, dim3((ncols + 63) / 64), dim3(256), 0, st, M, (long long)ld, mp, k, tau);
    }
    return hipGetLastError();
}

// In place on each right-hand side R[q] (row stride mp).
hipError_t launch_qr_solve(const double* M, int64_t ld, int mp, const double* tau, int nrhs, double* R,
                           int32_t* info, hipStream_t st) {
    if ((size_t)mp * sizeof(double) > 150 * 1024) return hipErrorInvalidValue;   // LDS-resident vector
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)qr_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           150 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    for (int q = 0; q < nrhs; ++q)
        hipLaunchKernelGGL(qr_solve_kernel, dim3(1), dim3(1024), (size_t)mp * sizeof(double), st, M, (long long)ld, mp,
                           tau, R + (size_t)q * mp, info);
    return hipGetLastError();
}

}  // namespace lpipm
