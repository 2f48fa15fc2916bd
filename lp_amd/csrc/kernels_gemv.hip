// kernels_gemv.hip -- passes over A for the residual / direction GEMVs (HBM-bound).
//
//   gemv_n : y = add + A.w      feasible_point.rs:122 (r_P), newton_equations.rs:220 (sym_solve r),
//                               residual.rs:23
//   gemv_t : u = A^T.v          feasible_point.rs:123 (r_D), newton_equations.rs:223 (sym_solve u),
//                               residual.rs:25
// Both take 1 or 2 vectors per pass: the predictor's two sym_solve calls (newton_equations.rs:
// 187-188) read A once instead of twice.  Coalescing: every wave instruction reads 64 lanes x 16 B
// of ONE row of the row-major A.  Reductions are fixed-order (shuffle butterfly / split slabs
// summed in index order) so results are bitwise reproducible run to run.
#include "lpipm_internal.hpp"

namespace lpipm {

typedef double d2 __attribute__((ext_vector_type(2)));

// rows per wave: 2 for the passes over A (more bytes in flight per wave), 1 for short matrices (the blocks of the
// triangular solves: 1024 rows would otherwise occupy only half of the CUs)

// One wave computes GN_ROWS_PER_WAVE row dot products; np (padded row length) is a multiple of 16,
// W is zero beyond the true n, so the 128-wide strides need a tail guard only on np.
template <int NRHS, int GN_ROWS_PER_WAVE>
__global__ __launch_bounds__(256) void gemv_n_kernel(const double* __restrict__ A, long long lda, int m,
                                                     int np, const double* __restrict__ W, long long ldw,
                                                     const double* add0,
                                                     const double* add1,
                                                     double* Y, long long ldy, double alpha, BatchK bk) {
    if (batch_done(bk)) return;
    A = batch_ptr(A, bk); W = batch_ptr(W, bk); add0 = batch_ptr(add0, bk); add1 = batch_ptr(add1, bk);
    Y = batch_ptr(Y, bk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int GN_ROWS_PER_WG = 4 * GN_ROWS_PER_WAVE;
    const int row0 = blockIdx.x * GN_ROWS_PER_WG + wave * GN_ROWS_PER_WAVE;
    if (row0 >= m) return;
    double acc[GN_ROWS_PER_WAVE][NRHS];
#pragma unroll
    for (int r = 0; r < GN_ROWS_PER_WAVE; ++r)
#pragma unroll
        for (int q = 0; q < NRHS; ++q) acc[r][q] = 0.0;
    const double* a0 = A + (long long)row0 * lda;
    for (int k = 2 * lane; k < np; k += 128) {
        d2 wv[NRHS];
#pragma unroll
        for (int q = 0; q < NRHS; ++q) wv[q] = *(const d2*)(W + (long long)q * ldw + k);
#pragma unroll
        for (int r = 0; r < GN_ROWS_PER_WAVE; ++r) {
            // rows past m (only in the last workgroup) re-read row m-1; their result is not stored
            const int rr = row0 + r < m ? r : 0;
            const d2 av = *(const d2*)(a0 + (long long)rr * lda + k);
#pragma unroll
            for (int q = 0; q < NRHS; ++q) acc[r][q] += av[0] * wv[q][0] + av[1] * wv[q][1];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int r = 0; r < GN_ROWS_PER_WAVE; ++r)
#pragma unroll
            for (int q = 0; q < NRHS; ++q) acc[r][q] += __shfl_xor(acc[r][q], off, 64);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < GN_ROWS_PER_WAVE; ++r) {
            if (row0 + r >= m) continue;
#pragma unroll
            for (int q = 0; q < NRHS; ++q) {
                const double* add = q == 0 ? add0 : add1;
                const double base = add ? add[row0 + r] : 0.0;
                Y[(long long)q * ldy + row0 + r] = base + alpha * acc[r][q];
            }
        }
    }
}

// grid = (np / 512 rounded up, mp / GEMVT_ROWS).  Each thread owns two adjacent columns and walks
// GEMVT_ROWS rows; the row split s = blockIdx.y writes its slab Upart[s][q][:].
template <int NRHS>
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* __restrict__ A, long long lda, int np,
                                                     const double* __restrict__ V, long long ldv,
                                                     double* __restrict__ Upart, long long slab, BatchK bk) {
    if (batch_done(bk)) return;
    A = batch_ptr(A, bk); V = batch_ptr(V, bk); Upart = batch_ptr(Upart, bk);
    __shared__ double vs[NRHS][GEMVT_ROWS];
    const int tid = threadIdx.x;
    const int col = (blockIdx.x * 256 + tid) * 2;
    const int row0 = blockIdx.y * GEMVT_ROWS;
    for (int e = tid; e < NRHS * GEMVT_ROWS; e += 256)
        vs[e / GEMVT_ROWS][e % GEMVT_ROWS] = V[(long long)(e / GEMVT_ROWS) * ldv + row0 + e % GEMVT_ROWS];
    __syncthreads();
    if (col >= np) return;
    d2 acc[NRHS];
#pragma unroll
    for (int q = 0; q < NRHS; ++q) acc[q] = (d2){0.0, 0.0};
    const double* ap = A + (long long)row0 * lda + col;
#pragma unroll 8
    for (int r = 0; r < GEMVT_ROWS; ++r) {
        const d2 av = *(const d2*)(ap + (long long)r * lda);
#pragma unroll
        for (int q = 0; q < NRHS; ++q) acc[q] += av * vs[q][r];
    }
#pragma unroll
    for (int q = 0; q < NRHS; ++q)
        *(d2*)(Upart + ((long long)blockIdx.y * NRHS + q) * slab + col) = acc[q];
}

__global__ __launch_bounds__(256) void gemv_t_reduce_kernel(const double* __restrict__ Upart, int nsplit,
                                                            int nrhs, int np, double* __restrict__ U,
                                                            long long ldu) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= np) return;
    for (int q = 0; q < nrhs; ++q) {
        double s = 0.0;
        for (int sp = 0; sp < nsplit; ++sp) s += Upart[((long long)sp * nrhs + q) * np + k];
        U[(long long)q * ldu + k] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// gemv_dual: A.w AND A^T.v in ONE read of A -- the residual pair of Residuals::calculate (residual.rs:23,25), which is
// also the r_P, r_D of the next get_delta (feasible_point.rs:122-123).
//   unit (row block of 128 rows, chunk of CW columns); wave w takes rows w, w+4, ...; a lane holds 2 adjacent columns
//   per 128-column step:
//     row part     AxPart[ch][r]  = sum_{c in chunk} A(r,c) w[c]      (wave reduction per row)
//     column part  Upart[rb][c]   = sum_{r in block} A(r,c) v[r]      (lane accumulators, the 4 waves added in order)
//   the consumers (k_residuals) add the chunk slabs / the row-block slabs in index order.
template <int CW>
__global__ __launch_bounds__(256) void gemv_dual_kernel(const double* __restrict__ A, long long lda, int np,
                                                        const double* __restrict__ W, const double* __restrict__ V,
                                                        double* __restrict__ AxPart, long long mp,
                                                        double* __restrict__ Upart, long long slab, BatchK bk) {
    if (batch_done(bk)) return;
    A = batch_ptr(A, bk); W = batch_ptr(W, bk); V = batch_ptr(V, bk); AxPart = batch_ptr(AxPart, bk); Upart = batch_ptr(Upart, bk);
    constexpr int NS = CW / 128;
    const int ch = blockIdx.x, rb = blockIdx.y;
    const int r0 = rb * GEMVT_ROWS, c0 = ch * CW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double vrow[GEMVT_ROWS];
    __shared__ double csum[4][CW];
    for (int e = threadIdx.x; e < GEMVT_ROWS; e += 256) vrow[e] = V[r0 + e];
    d2 wc[NS], cacc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int col = c0 + s * 128 + 2 * lane;
        wc[s] = col < np ? *(const d2*)(W + col) : (d2){0.0, 0.0};
        cacc[s] = (d2){0.0, 0.0};
    }
    __syncthreads();
    // RG rows per trip: their loads are all issued before any is used (RG * NS 16-byte loads in flight per lane) and
    // the RG row sums go through the shuffle butterfly together
    constexpr int RG = NS >= 8 ? 2 : 4;
    for (int rr0 = wave * RG; rr0 < GEMVT_ROWS; rr0 += 4 * RG) {
        d2 a[RG][NS];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const double* row = A + (long long)(r0 + rr0 + g) * lda + c0 + 2 * lane;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                a[g][s] = c0 + s * 128 + 2 * lane < np ? *(const d2*)(row + s * 128) : (d2){0.0, 0.0};
        }
        double racc[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const double vr = vrow[rr0 + g];
            racc[g] = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                racc[g] += a[g][s][0] * wc[s][0] + a[g][s][1] * wc[s][1];
                cacc[s] += a[g][s] * vr;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int g = 0; g < RG; ++g) racc[g] += __shfl_xor(racc[g], off, 64);
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < RG; ++g) AxPart[(long long)ch * mp + r0 + rr0 + g] = racc[g];
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) *(d2*)&csum[wave][s * 128 + 2 * lane] = cacc[s];
    __syncthreads();
    for (int e = threadIdx.x; e < CW; e += 256)
        if (c0 + e < np) Upart[(long long)rb * slab + c0 + e] = ((csum[0][e] + csum[1][e]) + csum[2][e]) + csum[3][e];
}
int gemv_dual_chunks(int np) { return np >= 4096 ? (np + 1023) / 1024 : (np + 255) / 256; }
hipError_t launch_gemv_dual(const double* A, int64_t lda, int mp, int np, const double* W, const double* V, double* AxPart,
                            double* Upart, int64_t slab, hipStream_t st, const Batch& bt) {
    if (slab <= 0) slab = np;
    const dim3 grid(gemv_dual_chunks(np), mp / GEMVT_ROWS, bt.count);
    if (np >= 4096) hipLaunchKernelGGL(gemv_dual_kernel<1024>, grid, dim3(256), 0, st, A, (long long)lda, np, W, V, AxPart, (long long)mp, Upart, (long long)slab, batch_k(bt));
    else            hipLaunchKernelGGL(gemv_dual_kernel<256>, grid, dim3(256), 0, st, A, (long long)lda, np, W, V, AxPart, (long long)mp, Upart, (long long)slab, batch_k(bt));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// rho = r0 - M.v for a symmetric M of which only the LOWER triangle is stored (the image A.D.A^T leaves), ONE read
// of that triangle: a stored element M(r,c), c < r, serves row r (M(r,c).v_c) and, as M(c,r), row c (M(r,c).v_r).
// This is the residual of the iterative-refinement step that makes the Cholesky solve of the normal equations as
// accurate as the reference's substitution (newton_equations.rs:151-169; the factor is consumed through explicit
// inverses of its diagonal super-blocks here, which alone loses digits on the ill-conditioned M of the last
// iterations: DESIGN.md 3.3).
//   unit (bi, ch): rows [64 bi, 64 bi + 64) x columns [512 ch, 512 ch + 512) of the lower triangle; wave w takes rows
//   w, w+4, ...; a lane holds 2 adjacent columns per 128-column step:
//     row part     slabA[ch][q][r]  = sum_{c in chunk, c <= r} M(r,c) v_q[c]            (wave reduction per row)
//     column part  slabB[bi][q][c]  = sum_{r in block, r > c} M(r,c) v_q[r]             (lane accumulators, 4 waves added)
//   symv_fold: rho_q[i] = r0_q[i] - ( sum_ch slabA[ch][q][i] + sum_{bi >= i/64} slabB[bi][q][i] ), fixed order.
// The products are accumulated in DOUBLED precision (error-free product by fma, error-free sum; Ogita-Rump-Oishi
// "Dot2"): a residual taken in working precision carries a rounding error of the size of the backward error of a
// stable solve, so refining with it makes good solves worse (measured: C4 member 217, rho_mu after the last
// iteration 1.3e-9 unrefined / 3.0e-8 refined with a plain residual, super-block width 128).  The kernel is bound by
// the one read of the triangle; the extra flops are free.
constexpr int SY_ROWS = 64, SY_COLS = 512;
struct dd { double hi, lo; };
// (contraction off: `hi + a*b` fused into one fma would not be the sum the error terms below are taken of)
__device__ __forceinline__ void dd_add_prod(dd& s, double a, double b) {       // s += a*b, a*b taken exactly
#pragma clang fp contract(off)
    const double p = a * b, e = fma(a, b, -p);
    const double t = s.hi + p, bb = t - s.hi;
    s.lo += ((s.hi - (t - bb)) + (p - bb)) + e;
    s.hi = t;
}
__device__ __forceinline__ void dd_add(dd& s, double hi, double lo) {           // s += (hi + lo)
#pragma clang fp contract(off)
    const double t = s.hi + hi, bb = t - s.hi;
    s.lo += ((s.hi - (t - bb)) + (hi - bb)) + lo;
    s.hi = t;
}
__host__ __device__ inline int symv_units(int mp) { const int nb = mp / SY_ROWS; int u = 0; for (int bi = 0; bi < nb; ++bi) u += bi / 8 + 1; return u; }
// slabs: [index][q][hi|lo][mp]
template <int NRHS>
__global__ __launch_bounds__(256) void symv_lower_kernel(const double* __restrict__ M, long long ld, int mp,
                                                         const double* __restrict__ V, long long ldv,
                                                         double* __restrict__ slabA, double* __restrict__ slabB, BatchK bk) {
    if (batch_done(bk)) return;
    M = batch_ptr(M, bk); V = batch_ptr(V, bk); slabA = batch_ptr(slabA, bk); slabB = batch_ptr(slabB, bk);
    // unit -> (bi, ch): row blocks 8g .. 8g+7 have g+1 column chunks each
    int u = blockIdx.x, g = 0;
    while (4 * (g + 1) * (g + 2) <= u) ++g;
    u -= 4 * g * (g + 1);
    const int bi = 8 * g + u / (g + 1), ch = u % (g + 1);
    const int r0 = bi * SY_ROWS, c0 = ch * SY_COLS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double vrow[NRHS][SY_ROWS];
    __shared__ double csum[4][NRHS][2][SY_COLS];
    for (int e = threadIdx.x; e < NRHS * SY_ROWS; e += 256) vrow[e / SY_ROWS][e % SY_ROWS] = V[(long long)(e / SY_ROWS) * ldv + r0 + e % SY_ROWS];
    d2 vc[SY_COLS / 128][NRHS];
    dd cacc[SY_COLS / 128][NRHS][2];
#pragma unroll
    for (int s = 0; s < SY_COLS / 128; ++s)
#pragma unroll
        for (int q = 0; q < NRHS; ++q) {
            const int col = c0 + s * 128 + 2 * lane;
            vc[s][q] = col < mp ? *(const d2*)(V + (long long)q * ldv + col) : (d2){0.0, 0.0};
            cacc[s][q][0] = dd{0.0, 0.0}; cacc[s][q][1] = dd{0.0, 0.0};
        }
    __syncthreads();
    for (int rr = wave; rr < SY_ROWS; rr += 4) {
        const int r = r0 + rr;
        const double* row = M + (long long)r * ld + c0 + 2 * lane;
        dd racc[NRHS];
        double vr[NRHS];
#pragma unroll
        for (int q = 0; q < NRHS; ++q) { racc[q] = dd{0.0, 0.0}; vr[q] = vrow[q][rr]; }
#pragma unroll
        for (int s = 0; s < SY_COLS / 128; ++s) {
            const int col = c0 + s * 128 + 2 * lane;
            if (c0 + s * 128 > r) break;                        // wave-uniform: nothing of the lower triangle further right
            d2 a = (d2){0.0, 0.0};
            if (col <= r) a = *(const d2*)(row + s * 128);
            if (col + 1 > r) a[1] = 0.0;                        // above the diagonal: not part of the stored triangle
#pragma unroll
            for (int q = 0; q < NRHS; ++q) {
                dd_add_prod(racc[q], a[0], vc[s][q][0]);
                dd_add_prod(racc[q], a[1], vc[s][q][1]);
                // the diagonal element belongs to the row part only
                dd_add_prod(cacc[s][q][0], col == r ? 0.0 : a[0], vr[q]);
                dd_add_prod(cacc[s][q][1], col + 1 == r ? 0.0 : a[1], vr[q]);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int q = 0; q < NRHS; ++q) {
                const double h = __shfl_xor(racc[q].hi, off, 64), l = __shfl_xor(racc[q].lo, off, 64);
                dd_add(racc[q], h, l);
            }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NRHS; ++q) {
                slabA[(((long long)ch * NRHS + q) * 2 + 0) * mp + r] = racc[q].hi;
                slabA[(((long long)ch * NRHS + q) * 2 + 1) * mp + r] = racc[q].lo;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < SY_COLS / 128; ++s)
#pragma unroll
        for (int q = 0; q < NRHS; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                csum[wave][q][0][s * 128 + 2 * lane + h] = cacc[s][q][h].hi;
                csum[wave][q][1][s * 128 + 2 * lane + h] = cacc[s][q][h].lo;
            }
    __syncthreads();
    for (int e = threadIdx.x; e < NRHS * SY_COLS; e += 256) {
        const int q = e / SY_COLS, c = e % SY_COLS;
        if (c0 + c >= mp) continue;
        dd t{csum[0][q][0][c], csum[0][q][1][c]};
#pragma unroll
        for (int w = 1; w < 4; ++w) dd_add(t, csum[w][q][0][c], csum[w][q][1][c]);
        slabB[(((long long)bi * NRHS + q) * 2 + 0) * mp + c0 + c] = t.hi;
        slabB[(((long long)bi * NRHS + q) * 2 + 1) * mp + c0 + c] = t.lo;
    }
}
template <int NRHS>
__global__ __launch_bounds__(256) void symv_fold_kernel(const double* __restrict__ R0, long long ldr, int mp,
                                                        const double* __restrict__ slabA, const double* __restrict__ slabB,
                                                        double* __restrict__ Rho, long long ldo, BatchK bk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= mp || batch_done(bk)) return;
    R0 = batch_ptr(R0, bk); slabA = batch_ptr(slabA, bk); slabB = batch_ptr(slabB, bk); Rho = batch_ptr(Rho, bk);
    const int nb = mp / SY_ROWS, bi0 = i / SY_ROWS, nch = (bi0 * SY_ROWS + SY_ROWS + SY_COLS - 1) / SY_COLS;
#pragma unroll
    for (int q = 0; q < NRHS; ++q) {
        dd t{0.0, 0.0};
        for (int ch = 0; ch < nch; ++ch)
            dd_add(t, slabA[(((long long)ch * NRHS + q) * 2 + 0) * mp + i], slabA[(((long long)ch * NRHS + q) * 2 + 1) * mp + i]);
        // column parts: every row block at or below row i (its 512-column chunk grid reaches column i)
        for (int bi = bi0; bi < nb; ++bi)
            dd_add(t, slabB[(((long long)bi * NRHS + q) * 2 + 0) * mp + i], slabB[(((long long)bi * NRHS + q) * 2 + 1) * mp + i]);
        dd r{R0[(long long)q * ldr + i], 0.0};
        dd_add(r, -t.hi, -t.lo);
        Rho[(long long)q * ldo + i] = r.hi + r.lo;
    }
}
// slab space (doubles) for an mp x mp symmetric product with up to 2 vectors (hi and lo parts)
size_t symv_slab_doubles(int mp) { return (size_t)((mp + SY_COLS - 1) / SY_COLS + mp / SY_ROWS) * 4 * mp; }
hipError_t launch_symv_residual(const double* M, int64_t ld, int mp, int nrhs, const double* V, int64_t ldv, const double* R0,
                                int64_t ldr, double* Rho, int64_t ldo, double* slabs, hipStream_t st, const Batch& bt) {
    double* slabA = slabs;
    double* slabB = slabs + (size_t)((mp + SY_COLS - 1) / SY_COLS) * 4 * mp;
    const dim3 grid(symv_units(mp), 1, bt.count), fgrid((mp + 255) / 256, 1, bt.count);
    if (nrhs == 1) {
        hipLaunchKernelGGL(symv_lower_kernel<1>, grid, dim3(256), 0, st, M, (long long)ld, mp, V, (long long)ldv, slabA, slabB, batch_k(bt));
        hipLaunchKernelGGL(symv_fold_kernel<1>, fgrid, dim3(256), 0, st, R0, (long long)ldr, mp, slabA, slabB, Rho, (long long)ldo, batch_k(bt));
    } else {
        hipLaunchKernelGGL(symv_lower_kernel<2>, grid, dim3(256), 0, st, M, (long long)ld, mp, V, (long long)ldv, slabA, slabB, batch_k(bt));
        hipLaunchKernelGGL(symv_fold_kernel<2>, fgrid, dim3(256), 0, st, R0, (long long)ldr, mp, slabA, slabB, Rho, (long long)ldo, batch_k(bt));
    }
    return hipGetLastError();
}

hipError_t launch_gemv_n(const double* A, int64_t lda, int m, int np, int nrhs, const double* W,
                         int64_t ldw, const double* add0, const double* add1, double* Y, int64_t ldy,
                         hipStream_t st, double alpha, const Batch& bt) {
    // rows per wave (a row's sum is the same whatever the count): 1 while the launch would otherwise leave CUs empty, 2 from
    // 2048 rows over the whole batch (every wave re-reads W from L2; C4 lockstep: solves 0.199 -> 0.187 ms, passes 0.527 ->
    // 0.521 per iteration; 4 rows per wave: the same)
    const int rpw = (long long)m * bt.count <= 2048 ? 1 : 2;
    const dim3 grid((m + 4 * rpw - 1) / (4 * rpw), 1, bt.count);
#define GN_LAUNCH(NR, RPW) hipLaunchKernelGGL((gemv_n_kernel<NR, RPW>), grid, dim3(256), 0, st, A, (long long)lda, m, np, W, \
                                              (long long)ldw, add0, add1, Y, (long long)ldy, alpha, batch_k(bt))
    if (nrhs == 1) { if (rpw == 1) GN_LAUNCH(1, 1); else GN_LAUNCH(1, 2); }
    else           { if (rpw == 1) GN_LAUNCH(2, 1); else GN_LAUNCH(2, 2); }
#undef GN_LAUNCH
    return hipGetLastError();
}

hipError_t launch_gemv_t(const double* A, int64_t lda, int mp, int np, int nrhs, const double* V,
                         int64_t ldv, double* Upart, hipStream_t st, int64_t slab, const Batch& bt) {
    if (slab <= 0) slab = np;
    dim3 grid((np / 2 + 255) / 256, mp / GEMVT_ROWS, bt.count);
    if (nrhs == 1)
        hipLaunchKernelGGL(gemv_t_kernel<1>, grid, dim3(256), 0, st, A, (long long)lda, np, V, (long long)ldv, Upart, (long long)slab, batch_k(bt));
    else
        hipLaunchKernelGGL(gemv_t_kernel<2>, grid, dim3(256), 0, st, A, (long long)lda, np, V, (long long)ldv, Upart, (long long)slab, batch_k(bt));
    return hipGetLastError();
}

// ---- slack structure (linear_program.rs:145-156): the last ns columns of the slack-form matrix are
// [I; 0] and are never stored; their contribution to the three matrix operations is added here.
//   A.w      : y[i]        += w[nx + i]          (i < ns)
//   A^T.v    : u[nx + i]    = v[i]               (goes to row-split slab 0; the other slabs hold 0 there)
//   A.D.A^T  : M[i][i]     += d[nx + i]
__global__ __launch_bounds__(256) void slack_n_kernel(int ns, int nx, int nrhs, const double* __restrict__ W,
                                                      long long ldw, double* __restrict__ Y, long long ldy, BatchK bk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns || batch_done(bk)) return;
    W = batch_ptr(W, bk); Y = batch_ptr(Y, bk);
    for (int q = 0; q < nrhs; ++q) Y[q * ldy + i] += W[q * ldw + nx + i];
}
// grid.y = row splits: split 0 receives v, every other split an explicit 0 (the slab buffer is shared by
// the 1- and 2-vector layouts, so "never written" is not the same as zero)
__global__ __launch_bounds__(256) void slack_t_kernel(int ns, int nx, int nrhs, const double* __restrict__ V,
                                                      long long ldv, double* __restrict__ Upart, long long slab, BatchK bk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns || batch_done(bk)) return;
    V = batch_ptr(V, bk); Upart = batch_ptr(Upart, bk);
    const int sp = blockIdx.y;
    for (int q = 0; q < nrhs; ++q)
        Upart[((long long)sp * nrhs + q) * slab + nx + i] = sp == 0 ? V[q * ldv + i] : 0.0;
}
__global__ __launch_bounds__(256) void slack_diag_kernel(int ns, int nx, const double* __restrict__ d,
                                                         double* __restrict__ M, long long ldm, BatchK bk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns || batch_done(bk)) return;
    d = batch_ptr(d, bk); M = batch_ptr(M, bk);
    M[(long long)i * ldm + i] += d[nx + i];
}
hipError_t launch_slack_n(int ns, int nx, int nrhs, const double* W, int64_t ldw, double* Y, int64_t ldy, hipStream_t st,
                          const Batch& bt) {
    if (ns <= 0) return hipSuccess;
    hipLaunchKernelGGL(slack_n_kernel, dim3((ns + 255) / 256, 1, bt.count), dim3(256), 0, st, ns, nx, nrhs, W, (long long)ldw, Y,
                       (long long)ldy, batch_k(bt));
    return hipGetLastError();
}
hipError_t launch_slack_t(int ns, int nx, int nrhs, int nsplit, const double* V, int64_t ldv, double* Upart, int64_t slab,
                          hipStream_t st, const Batch& bt) {
    if (ns <= 0) return hipSuccess;
    hipLaunchKernelGGL(slack_t_kernel, dim3((ns + 255) / 256, nsplit, bt.count), dim3(256), 0, st, ns, nx, nrhs, V, (long long)ldv,
                       Upart, (long long)slab, batch_k(bt));
    return hipGetLastError();
}
hipError_t launch_slack_diag(int ns, int nx, const double* d, double* M, int64_t ldm, hipStream_t st, const Batch& bt) {
    if (ns <= 0) return hipSuccess;
    hipLaunchKernelGGL(slack_diag_kernel, dim3((ns + 255) / 256, 1, bt.count), dim3(256), 0, st, ns, nx, d, M, (long long)ldm,
                       batch_k(bt));
    return hipGetLastError();
}

hipError_t launch_gemv_t_reduce(const double* Upart, int nsplit, int nrhs, int np, double* U, int64_t ldu,
                                hipStream_t st) {
    hipLaunchKernelGGL(gemv_t_reduce_kernel, dim3((np + 255) / 256), dim3(256), 0, st, Upart, nsplit, nrhs,
                       np, U, (long long)ldu);
    return hipGetLastError();
}

}  // namespace lpipm
