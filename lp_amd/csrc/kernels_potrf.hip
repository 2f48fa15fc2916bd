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
constexpr int DT  = 512;       // threads of the diagonal-block kernel (8 waves, 2 per SIMD)

typedef double d2 __attribute__((ext_vector_type(2)));

typedef double d4 __attribute__((ext_vector_type(4)));

// 1/sqrt(d): hardware seed (v_rsq_f64) + two Newton steps; NaN for d <= 0 propagates to L.
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-h * y, y, 0.5);
        y = fma(y, e, y);
    }
    return y;
}
// 1/d: hardware seed (v_rcp_f64) + two Newton steps.
__device__ __forceinline__ double rcp_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-d, r, 1.0);
        r = fma(r, e, r);
    }
    return r;
}
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

// 16x16 MFMA tiles on the LDS image (v_mfma_f64_16x16x4_f64: A lane l = A[l&15][k=l>>4],
// B lane l = B[k=l>>4][n=l&15], C/D reg r = C[(l>>4)+4r][l&15]).  With row stride 130 the
// A-form fragment read (16 rows x 2 k per 32-lane LDS phase) is bank-conflict free.
//   acc += sign * sum_{k<4*ksteps} X[ar+i][ac+k] * Y[br+j][bc+k]          ("NT": both K-contiguous)
__device__ __forceinline__ d4 mfma_nt16(d4 acc, const double (*Ls)[LS], int ar, int ac, int br, int bc,
                                        int ksteps, double sign, int fr, int fq) {
    for (int q = 0; q < ksteps; ++q) {
        const double a = sign * Ls[ar + fr][ac + 4 * q + fq];
        const double b = Ls[br + fr][bc + 4 * q + fq];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
//   acc += sum_{k<4*ksteps} X[ar+i][ac+k] * Y[br+k][bc+j]                  ("NN")
__device__ __forceinline__ d4 mfma_nn16(d4 acc, const double (*Ls)[LS], int ar, int ac, int br, int bc,
                                        int ksteps, int fr, int fq) {
    for (int q = 0; q < ksteps; ++q) {
        const double a = Ls[ar + fr][ac + 4 * q + fq];
        const double b = Ls[br + 4 * q + fq][bc + fr];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ void store_tile16(double (*Ls)[LS], int r0, int c0, d4 v, double sign, int fr, int fq) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Ls[r0 + fq + 4 * r][c0 + fr] = sign * v[r];
}

// One doubling level of the triangular inverse, block size S -> 2S, all pairs at once:
//   Inv21 = -Inv22 . (L21 . Inv11)   (16x16 MFMA tiles; zero blocks of the triangular factors
//   are skipped; the result replaces L21 in place, so each product is computed into registers,
//   then written after a barrier).
template <int S>
__device__ __forceinline__ void trtri_level(double (*Ls)[LS], int wave, int fr, int fq) {
    constexpr int NPAIR = NB / (2 * S);
    constexpr int TB = S / SB;                  // 16-tiles per block edge
    constexpr int NT = NPAIR * TB * TB;         // output tiles of the level
    constexpr int PER = (NT + 7) / 8;           // tiles per wave (8 waves)
    d4 acc[PER];
    // phase 1: T = L21 . Inv11  -> T(ti,tj) = sum_{tk >= tj} L21(ti,tk) Inv11(tk,tj)
#pragma unroll
    for (int w = 0; w < PER; ++w) {
        const int t = wave + 8 * w;
        acc[w] = (d4){0.0, 0.0, 0.0, 0.0};
        if (t < NT) {
            const int pair = t / (TB * TB), ti = (t / TB) % TB, tj = t % TB;
            const int base = pair * 2 * S;
            acc[w] = mfma_nn16(acc[w], Ls, base + S + ti * SB, base + tj * SB, base + tj * SB, base + tj * SB,
                               4 * (TB - tj), fr, fq);
        }
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < PER; ++w) {
        const int t = wave + 8 * w;
        if (t < NT) {
            const int pair = t / (TB * TB), ti = (t / TB) % TB, tj = t % TB;
            const int base = pair * 2 * S;
            store_tile16(Ls, base + S + ti * SB, base + tj * SB, acc[w], 1.0, fr, fq);
        }
    }
    __syncthreads();
    // phase 2: Inv21 = -Inv22 . T  -> (ti,tj) = -sum_{tk <= ti} Inv22(ti,tk) T(tk,tj)
#pragma unroll
    for (int w = 0; w < PER; ++w) {
        const int t = wave + 8 * w;
        acc[w] = (d4){0.0, 0.0, 0.0, 0.0};
        if (t < NT) {
            const int pair = t / (TB * TB), ti = (t / TB) % TB, tj = t % TB;
            const int base = pair * 2 * S;
            acc[w] = mfma_nn16(acc[w], Ls, base + S + ti * SB, base + S, base + S, base + tj * SB,
                               4 * (ti + 1), fr, fq);
        }
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < PER; ++w) {
        const int t = wave + 8 * w;
        if (t < NT) {
            const int pair = t / (TB * TB), ti = (t / TB) % TB, tj = t % TB;
            const int base = pair * 2 * S;
            store_tile16(Ls, base + S + ti * SB, base + tj * SB, acc[w], -1.0, fr, fq);
        }
    }
    __syncthreads();
}

// Mblk: top-left of the 128x128 diagonal block (row-major, ld).  Linv / LinvT: where inv(L_kk) and
// its transpose go (row-major, ldinv): the diagonal 128-blocks of the super-block inverses.
//
// Factorisation: 8 block columns of 16.  For each, every wave eliminates the 16 columns on
// 64 rows held one per lane: lanes 0-15 replicate the 16 diagonal rows (so no wave waits on
// another), lanes 16-63 carry panel rows -- the panel's triangular solve is the same elimination
// and comes for free.  Per pivot the critical chain is: v_readlane of the pivot -> Newton
// reciprocal -> one fma on the next pivot column -> publish that column to LDS (read back as
// broadcast by every lane, double-buffered); the 1/sqrt scaling of the finished column is off
// the chain.  Then a rank-16 update of the trailing lower triangle as 16x16 MFMA tiles.
// Inverse: the eight 16x16 diagonal factors are inverted in parallel (one wave each), then three
// doubling levels (16->32->64->128) of Inv21 = -Inv22.L21.Inv11 on MFMA tiles.
__global__ __launch_bounds__(DT) void potrf_diag_kernel(double* __restrict__ Mblk, long long ld,
                                                        double* __restrict__ Linv, double* __restrict__ LinvT,
                                                        long long ldinv, int32_t* info, int global_row0,
                                                        long long* stamps) {
#define STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[i] = clock64(); } while (0)
    __shared__ __attribute__((aligned(16))) double Ls[NB][LS];            // 133,120 B
    __shared__ __attribute__((aligned(16))) double scr[DT / 64][2][SB];   // per-wave column scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    for (int e = tid * 2; e < NB * NB; e += 2 * DT) {
        const int r = e >> 7, c = e & 127;
        d2 v = *(const d2*)(Mblk + (long long)r * ld + c);
        if (c > r) v[0] = 0.0;
        if (c + 1 > r) v[1] = 0.0;
        *(d2*)&Ls[r][c] = v;
    }
    __syncthreads();
    STAMP(0);
    for (int jb = 0; jb < NSB; ++jb) {
        const int c0 = jb * SB;
        const int npanel = NB - c0 - SB;
        // ---- fused elimination of block column jb
        const int prow = c0 + SB + wave * 48 + (lane - SB);       // panel row of lanes 16..63
        const bool is_diag = lane < SB;
        const bool wave_has_rows = wave == 0 || wave * 48 < npanel;
        if (wave_has_rows) {
            const bool valid = is_diag || prow < NB;
            const int row = is_diag ? c0 + lane : (valid ? prow : NB - 1);
            double a[SB];
#pragma unroll
            for (int c = 0; c < SB; c += 2) {
                const d2 v = *(const d2*)&Ls[row][c0 + c];
                a[c] = v[0]; a[c + 1] = v[1];
            }
            // Chain-critical values travel by v_readlane: the pivot a[j] of lane j and the first two
            // multipliers (a[j] of lanes j+1, j+2).  The rest of column j goes through LDS: its
            // reads are issued at the top of step j and consumed at the bottom, after the
            // reciprocal chain, so the LDS round trip overlaps the chain (a wave issues in order).
            if (jb == 0) STAMP(8);
            double piv = readlane_f64(a[0], 0);
            double c1 = readlane_f64(a[0], 1), c2 = readlane_f64(a[0], 2);
            double pivs[SB];
            if (is_diag) scr[wave][0][lane] = a[0];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                __builtin_amdgcn_wave_barrier();
                double col[SB];
#pragma unroll
                for (int k = (j + 3) & ~1; k < SB; k += 2) {
                    const d2 v = *(const d2*)&scr[wave][j & 1][k];
                    col[k] = v[0]; col[k + 1] = v[1];
                }
                pivs[j] = piv;
                const double rinv = rcp_nr(piv);
                const double t = a[j] * rinv;
                double piv_next = 0.0, c1_next = 0.0, c2_next = 0.0;
                if (j + 1 < SB) {
                    a[j + 1] = fma(-t, c1, a[j + 1]);
                    piv_next = readlane_f64(a[j + 1], j + 1);
                    if (j + 2 < SB) c1_next = readlane_f64(a[j + 1], j + 2);
                    if (j + 3 < SB) c2_next = readlane_f64(a[j + 1], j + 3);
                    if (is_diag) scr[wave][(j + 1) & 1][lane] = a[j + 1];
                }
                if (j + 2 < SB) a[j + 2] = fma(-t, c2, a[j + 2]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = j + 3; k < SB; ++k) a[k] = fma(-t, col[k], a[k]);
#pragma unroll
                for (int k = j + 2; k < SB; ++k) asm volatile("" : "+v"(a[k]));   // keep the update eager
                piv = piv_next; c1 = c1_next; c2 = c2_next;
            }
            if (jb == 0) STAMP(9);
            // finished columns of L: scale by 1/sqrt(pivot) (off the chain); first bad pivot -> info
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                if (!(pivs[j] > 0.0) && wave == 0 && lane == 0) atomicCAS((int*)info, 0, global_row0 + c0 + j + 1);
                a[j] = a[j] * rsqrt_nr(pivs[j]);
            }
            if (jb == 0) STAMP(10);
            // diagonal rows: wave 0 writes (replicas are identical); strict upper part stays zero
            if (is_diag) {
                if (wave == 0) {
#pragma unroll
                    for (int c = 0; c < SB; ++c) Ls[row][c0 + c] = (c <= lane) ? a[c] : 0.0;
                }
            } else if (valid) {
#pragma unroll
                for (int c = 0; c < SB; c += 2) *(d2*)&Ls[row][c0 + c] = (d2){a[c], a[c + 1]};
            }
        }
        if (jb == 0) STAMP(11);
        __syncthreads();
        if (jb == 0) STAMP(1);
        if (npanel > 0) {
            // ---- rank-16 update of the trailing lower triangle, 16x16 MFMA tiles round-robin over waves
            const int nb16 = npanel / SB;
            const int ntile = nb16 * (nb16 + 1) / 2;
            const int base = c0 + SB;
            for (int t = wave; t < ntile; t += DT / 64) {
                int bi = 0;
                while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
                const int bk = t - bi * (bi + 1) / 2;
                const int i0 = base + SB * bi, k0 = base + SB * bk;
                d4 cacc;
#pragma unroll
                for (int r = 0; r < 4; ++r) cacc[r] = Ls[i0 + fq + 4 * r][k0 + fr];
                cacc = mfma_nt16(cacc, Ls, i0, c0, k0, c0, 4, -1.0, fr, fq);
                store_tile16(Ls, i0, k0, cacc, 1.0, fr, fq);
            }
            __syncthreads();
            if (jb == 0) STAMP(2);
        }
    }
    STAMP(3);
    STAMP(4);
    // ---- write L (lower triangle incl. diagonal) back; the strict upper triangle of M is untouched
    for (int e = tid * 2; e < NB * NB; e += 2 * DT) {
        const int r = e >> 7, c = e & 127;
        if (c + 1 <= r) *(d2*)(Mblk + (long long)r * ld + c) = *(const d2*)&Ls[r][c];
        else if (c <= r) Mblk[(long long)r * ld + c] = Ls[r][c];
    }
    STAMP(5);
    // ---- inverse of the eight 16x16 diagonal factors, one wave each (lanes 0-15 = rows of the
    // inverse): x.L16 = e_row solved right to left; L values are wave-uniform broadcast reads.
    {
        const int c0 = wave * SB, row = lane & 15;
        if (lane < SB) scr[wave][0][lane] = rcp_nr(Ls[c0 + lane][c0 + lane]);
        __builtin_amdgcn_wave_barrier();
        double x[SB];
#pragma unroll
        for (int j = SB - 1; j >= 0; --j) {
            double s = (row == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = j + 1; k < SB; ++k) s = fma(-x[k], Ls[c0 + k][c0 + j], s);
            x[j] = s * scr[wave][0][j];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < SB) {
#pragma unroll
            for (int c = 0; c < SB; ++c) Ls[c0 + row][c0 + c] = (c <= row) ? x[c] : 0.0;
        }
    }
    __syncthreads();
    trtri_level<16>(Ls, wave, fr, fq);
    trtri_level<32>(Ls, wave, fr, fq);
    trtri_level<64>(Ls, wave, fr, fq);
    STAMP(6);
    for (int e = tid * 2; e < NB * NB; e += 2 * DT) {
        const int r = e >> 7, c = e & 127;
        d2 v = *(const d2*)&Ls[r][c];
        if (c > r) v[0] = 0.0;
        if (c + 1 > r) v[1] = 0.0;
        *(d2*)(Linv + (long long)r * ldinv + c) = v;
        // transposed copy (upper triangular), same (r, c) walk so the global stores stay coalesced
        d2 w = (d2){Ls[c][r], Ls[c + 1][r]};
        if (r > c) w[0] = 0.0;
        if (r > c + 1) w[1] = 0.0;
        *(d2*)(LinvT + (long long)r * ldinv + c) = w;
    }
    STAMP(7);
#undef STAMP
}

long long* g_diag_stamps = nullptr;  // debug: device buffer of 8 cycle stamps for block 0 (scripts/)
hipError_t launch_potrf(double* M, int64_t ld, int mp, const FactorPlan& plan, int32_t* info, hipStream_t st) {
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    const int nb = mp / NB;
    for (int k = 0; k < nb; ++k) {
        const int64_t o = (int64_t)k * NB;
        double* diag = M + o * ld + o;
        double* linv = plan.blk_inv(k);
        const int ldinv = plan.blk_ld(k);
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(DT), 0, st, diag, (long long)ld, linv, plan.blk_invT(k),
                           (long long)ldinv, info, (int)o, (long long*)(k == 0 ? g_diag_stamps : nullptr));
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int rem = nb - k - 1;
        if (rem <= 0) break;
        double* panel = M + (o + NB) * ld + o;
        GemmArgs t{};
        t.P = panel; t.ldp = ld; t.Q = linv; t.ldq = ldinv; t.s = nullptr;
        t.C = panel; t.ldc = ld; t.K = NB; t.alpha = 1.0; t.beta = 0.0;
        // few tiles -> latency-bound: 32-row x 128-col tiles put 4x as many CUs on the panel, and a
        // workgroup still owns whole rows, so the product may overwrite its own input
        t.tile_edge = 32;
        t.ntiles = 4 * rem; t.tiles_lower = 0; t.ntj = 1; t.tile_list = nullptr;
        t.diag_pad_from = -1; t.ws = nullptr; t.nwg = t.ntiles;
        e = launch_gemm_nt(t, st);
        if (e != hipSuccess) return e;
        GemmArgs u{};
        u.P = panel; u.ldp = ld; u.Q = panel; u.ldq = ld; u.s = nullptr;
        u.C = M + (o + NB) * ld + (o + NB); u.ldc = ld; u.K = NB; u.alpha = -1.0; u.beta = 1.0;
        u.tiles_lower = 1; u.ntj = 0; u.tile_list = nullptr;
        if (rem * (rem + 1) / 2 < 256) { u.tile_edge = 64; u.ntiles = (2 * rem) * (2 * rem + 1) / 2; }
        else                           { u.tile_edge = 128; u.ntiles = rem * (rem + 1) / 2; }
        u.diag_pad_from = -1; u.ws = nullptr; u.nwg = u.ntiles;
        e = launch_gemm_nt(u, st);
        if (e != hipSuccess) return e;
    }
    // inverses of the diagonal super-blocks from the 128-block inverses: doubling levels, each a
    // grouped launch of stage A (T^T = Inv11^T.L21^T) then stage B (Inv21 = -Inv22.T and its transpose)
    for (const auto& stg : plan.stages) {
        e = launch_gemm_grouped(plan.descs_dev + stg.first, stg.second, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace lpipm
