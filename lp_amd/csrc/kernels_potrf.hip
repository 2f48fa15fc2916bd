// kernels_potrf.hip -- blocked lower Cholesky M = L.L^T on device (replaces `M.cholesky()`,
// newton_equations.rs:129-131; the reference's default backend is an unblocked scalar loop).
//
// Right-looking, block size NB = 128 (== the MFMA GEMM tile):
//   for each block column k:
//     1. potrf_diag_kernel  (ONE workgroup): factor the 128x128 diagonal block in LDS and form
//        inv(L_kk); both are needed downstream (inv(L_kk) turns the panel TRSM into a GEMM and the
//        triangular solves into block mat-vecs).
//     2. L21 = A21 . inv(L_kk)^T           -> MFMA NT-GEMM, in place
//     3. A22 -= L21 . L21^T (lower tiles)  -> MFMA NT-GEMM, alpha=-1, beta=1
// A non-positive pivot is recorded in `info` (1 + its global index, first one wins) and the
// factorisation continues on NaNs; the host maps info != 0 to NumericalProblem exactly where the
// reference maps a failed `cholesky()` (newton_equations.rs:59-63).
#include "lpipm_internal.hpp"

namespace lpipm {

constexpr int SB  = 16;        // sub-block edge inside the diagonal block
constexpr int NSB = NB / SB;   // 8
constexpr int LS  = NB + 2;    // LDS row stride in doubles (260 dwords: rows 4 banks apart)

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

// One wave factors the 16x16 diagonal sub-block at (c0,c0) of Ls and inverts the factor.
// Lane (l & 15) owns row l&15 in registers; cross-row values move by v_readlane (no LDS, no barrier).
__device__ __forceinline__ void potrf16_trtri16(double (*Ls)[LS], double (*I16)[SB + 1], int c0,
                                                int lane, int32_t* info, int global_row0) {
    const int row = lane & 15;
    double a[SB], x[SB];
#pragma unroll
    for (int c = 0; c < SB; ++c) a[c] = (c <= row) ? Ls[c0 + row][c0 + c] : 0.0;

#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const double d = readlane_f64(a[j], j);
        if (!(d > 0.0)) {  // wave-uniform: d comes from one lane
            if (lane == 0) atomicCAS((int*)info, 0, global_row0 + c0 + j + 1);
        }
        const double l = sqrt(d);
        a[j] = (row == j) ? l : a[j] / l;
#pragma unroll
        for (int k = j + 1; k < SB; ++k) {
            const double lkj = readlane_f64(a[j], k);
            a[k] = (row >= k) ? a[k] - a[j] * lkj : a[k];
        }
    }
    // row `row` of inv(L16): x . L16 = e_row, columns solved right to left
#pragma unroll
    for (int j = SB - 1; j >= 0; --j) {
        double s = (row == j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = j + 1; k < SB; ++k) s -= x[k] * readlane_f64(a[j], k);
        x[j] = s / readlane_f64(a[j], j);
    }
    if (lane < SB) {
#pragma unroll
        for (int c = 0; c < SB; ++c) {
            Ls[c0 + row][c0 + c] = (c <= row) ? a[c] : 0.0;
            I16[row][c] = (c <= row) ? x[c] : 0.0;
        }
    }
}

// Mblk: top-left of the diagonal block (row-major, ld).  Linv: 128x128 row-major slab.
__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ Mblk, long long ld,
                                                         double* __restrict__ Linv, int32_t* info,
                                                         int global_row0) {
    __shared__ __attribute__((aligned(16))) double Ls[NB][LS];       // 133,120 B
    __shared__ __attribute__((aligned(16))) double I16[NSB][SB][SB + 1];  // 17,408 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 7, c = e & 127;
        Ls[r][c] = (c <= r) ? Mblk[(long long)r * ld + c] : 0.0;
    }
    __syncthreads();

    for (int jb = 0; jb < NSB; ++jb) {
        const int c0 = jb * SB;
        if (wave == 0) potrf16_trtri16(Ls, I16[jb], c0, lane, info, global_row0);
        __syncthreads();
        const int nr = NB - c0 - SB;  // rows below the diagonal sub-block
        if (nr > 0) {
            // ---- panel: X = B . inv(L16)^T, two threads per row (8 columns each)
            const int pr = tid >> 1, half = tid & 1;
            double brow[SB];
            const bool act = pr < nr;
            if (act) {
#pragma unroll
                for (int k = 0; k < SB; ++k) brow[k] = Ls[c0 + SB + pr][c0 + k];
            }
            __syncthreads();
            if (act) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = half * 8 + jj;
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < SB; ++k)
                        if (k <= j) s += brow[k] * I16[jb][j][k];
                    Ls[c0 + SB + pr][c0 + j] = s;
                }
            }
            __syncthreads();
            // ---- trailing update of the lower triangle: 4x4 register tiles
            const int nt4 = nr >> 2;
            const int ntile = nt4 * (nt4 + 1) / 2;
            const int base = c0 + SB;
            for (int t = tid; t < ntile; t += 256) {
                int bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
                while (bi * (bi + 1) / 2 > t) --bi;
                const int bk = t - bi * (bi + 1) / 2;
                const int i0 = base + 4 * bi, k0 = base + 4 * bk;
                double accu[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) accu[u][v] = 0.0;
#pragma unroll 4
                for (int j = 0; j < SB; ++j) {
                    double xi[4], xk[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) xi[u] = Ls[i0 + u][c0 + j];
#pragma unroll
                    for (int v = 0; v < 4; ++v) xk[v] = Ls[k0 + v][c0 + j];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) accu[u][v] += xi[u] * xk[v];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (k0 + v <= i0 + u) Ls[i0 + u][k0 + v] -= accu[u][v];
            }
            __syncthreads();
        }
    }

    // ---- write L (lower triangle incl. diagonal) back; the strict upper triangle of M is untouched
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 7, c = e & 127;
        if (c <= r) Mblk[(long long)r * ld + c] = Ls[r][c];
    }
    __syncthreads();
    // diagonal sub-blocks <- their inverses; Ls now holds inv16 on the diagonal, L below it
    for (int e = tid; e < NSB * SB * SB; e += 256) {
        const int b = e >> 8, r = (e >> 4) & 15, c = e & 15;
        Ls[b * SB + r][b * SB + c] = I16[b][r][c];
    }
    __syncthreads();

    // ---- inv(L) in place, block column by block column from the right:
    //   Inv[i][j] = -( sum_{k=j+1..i} Inv[i][k] . L[k][j] ) . inv16_j        (i > j)
    for (int jb = NSB - 2; jb >= 0; --jb) {
        const int c0 = jb * SB;
        const int nr = NB - c0 - SB;
        const int pr = tid >> 1, half = tid & 1;
        const bool act = pr < nr;
        const int r = c0 + SB + pr;
        // (a) W = L[:, j] . inv16_j   (rows below the diagonal block), in place
        double brow[SB];
        if (act) {
#pragma unroll
            for (int k = 0; k < SB; ++k) brow[k] = Ls[r][c0 + k];
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int c = half * 8 + jj;
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < SB; ++k)
                    if (k >= c) s += brow[k] * Ls[c0 + k][c0 + c];  // inv16_j is lower triangular
                Ls[r][c0 + c] = s;
            }
        }
        __syncthreads();
        // (b) Inv[r][c0 + c] = - sum_{k = c0+16 .. end of r's sub-block} Inv[r][k] . W[k][c]
        double out[8];
        if (act) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) out[jj] = 0.0;
            const int kend = (r | (SB - 1));
            for (int k = c0 + SB; k <= kend; ++k) {
                const double ark = Ls[r][k];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) out[jj] -= ark * Ls[k][c0 + half * 8 + jj];
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) Ls[r][c0 + half * 8 + jj] = out[jj];
        }
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 7, c = e & 127;
        Linv[e] = (c <= r) ? Ls[r][c] : 0.0;
    }
}

// Marks a failed factorisation that produced NaN without tripping the pivot test (defensive).
hipError_t launch_potrf(double* M, int64_t ld, int mp, double* invL, int32_t* info, hipStream_t st) {
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    const int nb = mp / NB;
    for (int k = 0; k < nb; ++k) {
        const int64_t o = (int64_t)k * NB;
        double* diag = M + o * ld + o;
        double* linv = invL + (int64_t)k * NB * NB;
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, st, diag, (long long)ld, linv, info,
                           (int)o);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int rem = nb - k - 1;
        if (rem <= 0) break;
        double* panel = M + (o + NB) * ld + o;
        GemmArgs t{};
        t.P = panel; t.ldp = ld; t.Q = linv; t.ldq = NB; t.s = nullptr;
        t.C = panel; t.ldc = ld; t.K = NB; t.alpha = 1.0; t.beta = 0.0;
        t.ntiles = rem; t.tiles_lower = 0; t.ntj = 1; t.tile_list = nullptr;
        t.diag_pad_from = -1; t.ws = nullptr; t.nwg = rem;
        e = launch_gemm_nt(t, st);
        if (e != hipSuccess) return e;
        GemmArgs u{};
        u.P = panel; u.ldp = ld; u.Q = panel; u.ldq = ld; u.s = nullptr;
        u.C = M + (o + NB) * ld + (o + NB); u.ldc = ld; u.K = NB; u.alpha = -1.0; u.beta = 1.0;
        u.ntiles = rem * (rem + 1) / 2; u.tiles_lower = 1; u.ntj = 0; u.tile_list = nullptr;
        u.diag_pad_from = -1; u.ws = nullptr; u.nwg = u.ntiles;
        e = launch_gemm_nt(u, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace lpipm
