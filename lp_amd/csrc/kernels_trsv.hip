// kernels_trsv.hip -- v = L^-T (L^-1 r) with the blocked factor of kernels_potrf.hip
// (replaces `factor.solvec_into(b2)`, newton_equations.rs:151-169; called from sym_solve :221).
//
// Block right-looking substitution with NB = 128 and the explicit inverses of the diagonal blocks
// (a triangular solve inside a block becomes a dense 128x128 mat-vec, no sequential inner loop):
//   forward  (k = 0..nb-1):  y_k = inv(L_kk) . r_k ;   r_i -= L_ik . y_k      for i > k
//   backward (k = nb-1..0):  v_k = inv(L_kk)^T . y_k ; y_j -= L_kj^T . v_k    for j < k
// One launch per block step (the step's solve is recomputed by every workgroup of the launch, so
// a step is ONE kernel and needs no inter-workgroup hand-off); 1 or 2 right-hand sides per sweep
// (the predictor's two sym_solve calls, newton_equations.rs:187-188, share one sweep).
// HBM-bound: a sweep reads the lower triangle of L once (4 m^2 bytes) + the diagonal inverses.
#include "lpipm_internal.hpp"

namespace lpipm {

typedef double d2 __attribute__((ext_vector_type(2)));

// out[r] (+)= sum_c B[r][c] * x[c] for a dense 128x128 row-major block (ld), rows split over the 4
// waves (32 each); a wave reads one full row per instruction (64 lanes x 16 B) and butterfly-reduces.
// xs: x in LDS [nrhs][128].  result for row r delivered to lane 0 of the wave -> res(r, rhs, value).
template <int NRHS, typename F>
__device__ __forceinline__ void block_matvec_n(const double* __restrict__ B, long long ld,
                                               const double (*xs)[NB], int wave, int lane, F&& res) {
    d2 xv[NRHS];
#pragma unroll
    for (int q = 0; q < NRHS; ++q) xv[q] = *(const d2*)&xs[q][2 * lane];
#pragma unroll 4
    for (int rr = 0; rr < 32; ++rr) {
        const int r = wave * 32 + rr;
        const d2 bv = *(const d2*)(B + (long long)r * ld + 2 * lane);
        double acc[NRHS];
#pragma unroll
        for (int q = 0; q < NRHS; ++q) acc[q] = bv[0] * xv[q][0] + bv[1] * xv[q][1];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int q = 0; q < NRHS; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NRHS; ++q) res(r, q, acc[q]);
        }
    }
}

// part[wave][q][c] = sum_{r in wave's 32 rows} B[r][c] * x[q][r]  (transposed product, lanes over c)
template <int NRHS>
__device__ __forceinline__ void block_matvec_t(const double* __restrict__ B, long long ld,
                                               const double (*xs)[NB], int wave, int lane,
                                               double (*part)[NRHS][NB]) {
    d2 acc[NRHS];
#pragma unroll
    for (int q = 0; q < NRHS; ++q) acc[q] = (d2){0.0, 0.0};
#pragma unroll 4
    for (int rr = 0; rr < 32; ++rr) {
        const int r = wave * 32 + rr;
        const d2 bv = *(const d2*)(B + (long long)r * ld + 2 * lane);
#pragma unroll
        for (int q = 0; q < NRHS; ++q) acc[q] += bv * xs[q][r];
    }
#pragma unroll
    for (int q = 0; q < NRHS; ++q) *(d2*)&part[wave][q][2 * lane] = acc[q];
}

// Forward step k.  grid = nb - k: workgroup w handles block row i = k + w.
//   R : right-hand sides, updated in place for rows below block k   [nrhs][mp]
//   Y : forward solution                                             [nrhs][mp]
template <int NRHS>
__global__ __launch_bounds__(256) void trsv_fwd_step(const double* __restrict__ L, long long ld,
                                                     const double* __restrict__ invL, int mp, int k,
                                                     double* __restrict__ R, double* __restrict__ Y) {
    __shared__ __attribute__((aligned(16))) double bs[NRHS][NB];
    __shared__ __attribute__((aligned(16))) double ys[NRHS][NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = k + blockIdx.x;
    for (int e = tid; e < NRHS * NB; e += 256) bs[e / NB][e % NB] = R[(long long)(e / NB) * mp + k * NB + e % NB];
    __syncthreads();
    block_matvec_n<NRHS>(invL + (long long)k * NB * NB, NB, bs, wave, lane,
                         [&](int r, int q, double v) { ys[q][r] = v; });
    __syncthreads();
    if (i == k) {
        for (int e = tid; e < NRHS * NB; e += 256) Y[(long long)(e / NB) * mp + k * NB + e % NB] = ys[e / NB][e % NB];
    } else {
        const double* blk = L + (long long)i * NB * ld + (long long)k * NB;
        block_matvec_n<NRHS>(blk, ld, ys, wave, lane, [&](int r, int q, double v) {
            R[(long long)q * mp + i * NB + r] -= v;
        });
    }
}

// Backward step k.  grid = k + 1: workgroup w < k handles column block j = w, workgroup k writes v_k.
//   Y : forward solution, updated in place for blocks above k      [nrhs][mp]
//   V : final solution                                              [nrhs][mp]
template <int NRHS>
__global__ __launch_bounds__(256) void trsv_bwd_step(const double* __restrict__ L, long long ld,
                                                     const double* __restrict__ invL, int mp, int k,
                                                     double* __restrict__ Y, double* __restrict__ V) {
    __shared__ __attribute__((aligned(16))) double bs[NRHS][NB];
    __shared__ __attribute__((aligned(16))) double vs[NRHS][NB];
    __shared__ __attribute__((aligned(16))) double part[4][NRHS][NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x;
    for (int e = tid; e < NRHS * NB; e += 256) bs[e / NB][e % NB] = Y[(long long)(e / NB) * mp + k * NB + e % NB];
    __syncthreads();
    block_matvec_t<NRHS>(invL + (long long)k * NB * NB, NB, bs, wave, lane, part);
    __syncthreads();
    for (int e = tid; e < NRHS * NB; e += 256) {
        const int q = e / NB, c = e % NB;
        vs[q][c] = (part[0][q][c] + part[1][q][c]) + (part[2][q][c] + part[3][q][c]);
    }
    __syncthreads();
    if (j == k) {
        for (int e = tid; e < NRHS * NB; e += 256) V[(long long)(e / NB) * mp + k * NB + e % NB] = vs[e / NB][e % NB];
    } else {
        const double* blk = L + (long long)k * NB * ld + (long long)j * NB;
        block_matvec_t<NRHS>(blk, ld, vs, wave, lane, part);
        __syncthreads();
        for (int e = tid; e < NRHS * NB; e += 256) {
            const int q = e / NB, c = e % NB;
            Y[(long long)q * mp + j * NB + c] -= (part[0][q][c] + part[1][q][c]) + (part[2][q][c] + part[3][q][c]);
        }
    }
}

// R is consumed (overwritten: first as forward scratch, finally with the solution).
// Yscratch: nrhs x mp doubles of workspace.
hipError_t launch_chol_solve_ws(const double* L, int64_t ld, const double* invL, int mp, int nrhs,
                                double* R, double* Yscratch, hipStream_t st) {
    const int nb = mp / NB;
    for (int k = 0; k < nb; ++k) {
        if (nrhs == 1)
            hipLaunchKernelGGL(trsv_fwd_step<1>, dim3(nb - k), dim3(256), 0, st, L, (long long)ld, invL, mp, k, R, Yscratch);
        else
            hipLaunchKernelGGL(trsv_fwd_step<2>, dim3(nb - k), dim3(256), 0, st, L, (long long)ld, invL, mp, k, R, Yscratch);
    }
    for (int k = nb - 1; k >= 0; --k) {
        if (nrhs == 1)
            hipLaunchKernelGGL(trsv_bwd_step<1>, dim3(k + 1), dim3(256), 0, st, L, (long long)ld, invL, mp, k, Yscratch, R);
        else
            hipLaunchKernelGGL(trsv_bwd_step<2>, dim3(k + 1), dim3(256), 0, st, L, (long long)ld, invL, mp, k, Yscratch, R);
    }
    return hipGetLastError();
}

}  // namespace lpipm
