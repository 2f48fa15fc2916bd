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
constexpr int DT  = 1024;      // threads of the diagonal-block kernel: 16 waves (3 eliminate, 13 update), 4 per SIMD

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
// Mblk: top-left of the 128x128 diagonal block (row-major, ld).  Linv / LinvT: where inv(L_kk) and
// its transpose go (row-major, ldinv): the diagonal 128-blocks of the super-block inverses.
//
// ONE elimination produces both the factor and its inverse.  Right-looking Cholesky applied to the
// augmented matrix [A; I] leaves [L; L^-T]: the extra identity rows are just more panel rows of the
// triangular solve X.L^T = B with B = I.  Row i of L^-T is zero left of column i, so its entries live
// in the (otherwise unused) strict upper triangle of the LDS image; the diagonal 1/L_ii goes to dinv.
//
// 8 block columns of 16.  For each, three waves eliminate the 16 columns on 144 rows held one per
// lane: lanes 0-15 of every wave replicate the 16 diagonal rows (no wave waits on another), lanes
// 16-63 carry the 128 other rows (panel rows below, identity rows above / inside the block).  Per
// pivot the critical chain is: v_readlane of pivot + two multipliers -> Newton reciprocal -> fma on
// the next pivot column; the rest of the column is broadcast through LDS one step later and the
// 1/sqrt scaling of finished columns is off the chain.  Then one rank-16 update of everything to
// the right (lower tiles of A and the identity rows' tiles in the upper triangle) as 16x16 MFMA
// tiles, two independent tiles in flight per wave.
// Barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for
// the fire-and-forget global stores of finished L columns (thousands of cycles per block column).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int XS = SB + 2;   // row stride of the compact identity-row operand (conflict-free MFMA reads)

// Eliminates block column jb (16 columns starting at c0) on 144 rows, one per lane of waves 0..2.
// xidb: where the eliminated NEW identity rows go as an MFMA operand.
__device__ __forceinline__ void eliminate_block_column(double (*Ls)[LS], double (*xidb)[XS], double (*scrw)[SB],
                                                       double* dinv, int jb, int wave, int lane, int32_t* info,
                                                       int global_row0, double* __restrict__ Mblk, long long ld) {
    const int c0 = jb * SB;
    const int npanel = NB - c0 - SB;                        // rows of A below the diagonal sub-block
    const bool is_diag = lane < SB;
    // non-diagonal rows: v in [0,128): panel rows first, then identity rows 0 .. c0+15
    const int v = wave * 48 + (lane - SB);
    const bool valid = is_diag || v < NB;
    const bool is_panel = !is_diag && v < npanel;
    const int idrow = v - npanel;                           // identity row index (when !is_panel)
    const bool is_newid = !is_diag && !is_panel && idrow >= c0;
    const int row = is_diag ? c0 + lane : (is_panel ? c0 + SB + v : (valid ? idrow : 0));
    double a[SB];
#pragma unroll
    for (int c = 0; c < SB; c += 2) {
        const d2 x = *(const d2*)&Ls[row][c0 + c];
        a[c] = x[0]; a[c + 1] = x[1];
    }
    if (is_newid) {                                         // e_i restricted to this block's columns
#pragma unroll
        for (int c = 0; c < SB; ++c) a[c] = (c == idrow - c0) ? 1.0 : 0.0;
    }
    // Chain-critical values travel by v_readlane: the pivot a[j] of lane j and the first two
    // multipliers (a[j] of lanes j+1, j+2).  The rest of column j goes through LDS: its reads are
    // issued at the top of step j and consumed at the bottom, after the reciprocal chain, so the
    // LDS round trip overlaps the chain (a wave issues in order).
    double piv = readlane_f64(a[0], 0);
    double c1 = readlane_f64(a[0], 1), c2 = readlane_f64(a[0], 2);
    if (is_diag) scrw[0][lane] = a[0];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        __builtin_amdgcn_wave_barrier();
        double col[SB];
#pragma unroll
        for (int k = (j + 3) & ~1; k < SB; k += 2) {
            const d2 x = *(const d2*)&scrw[j & 1][k];
            col[k] = x[0]; col[k + 1] = x[1];
        }
        const double rinv = rcp_nr(piv);
        const double t = a[j] * rinv;
        double piv_next = 0.0, c1_next = 0.0, c2_next = 0.0;
        if (j + 1 < SB) {
            a[j + 1] = fma(-t, c1, a[j + 1]);
            piv_next = readlane_f64(a[j + 1], j + 1);
            if (j + 2 < SB) c1_next = readlane_f64(a[j + 1], j + 2);
            if (j + 3 < SB) c2_next = readlane_f64(a[j + 1], j + 3);
            if (is_diag) scrw[(j + 1) & 1][lane] = a[j + 1];
        }
        if (j + 2 < SB) a[j + 2] = fma(-t, c2, a[j + 2]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = j + 3; k < SB; ++k) a[k] = fma(-t, col[k], a[k]);
#pragma unroll
        for (int k = j + 2; k < SB; ++k) asm volatile("" : "+v"(a[k]));   // keep the update eager
        piv = piv_next; c1 = c1_next; c2 = c2_next;
    }
    // finished columns: scale column j by 1/sqrt(pivot j).  Lane j of the diagonal rows still holds
    // pivot j in a[j]: it forms the rsqrt, all lanes pick the 16 values up from LDS.
    {
        double mine = a[0];
#pragma unroll
        for (int j = 1; j < SB; ++j) mine = (lane == j) ? a[j] : mine;
        const unsigned long long bad = __ballot(is_diag && !(mine > 0.0));
        if (bad != 0ull && wave == 0 && lane == 0)   // lowest set bit = first bad pivot of this block column
            atomicCAS((int*)info, 0, global_row0 + c0 + (int)__ffsll((long long)bad));
        const double myrs = rsqrt_nr(mine);
        __builtin_amdgcn_wave_barrier();
        if (is_diag) {
            scrw[0][lane] = myrs;
            if (wave == 0) dinv[c0 + lane] = myrs;          // 1/L_ii = 1/sqrt(pivot)
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < SB; j += 2) {
            const d2 rs = *(const d2*)&scrw[0][j];
            a[j] *= rs[0]; a[j + 1] *= rs[1];
        }
        __builtin_amdgcn_wave_barrier();
    }
    // write back.  Diagonal rows (wave 0 only; replicas are identical): nothing reads the factored
    // diagonal sub-block from LDS again, so it goes straight to global memory -- and the LDS copy
    // stays as loaded, which keeps the other waves' replica loads race-free.
    // Panel / old identity rows: all 16 columns.  New identity rows: row q of inv(L16)^T goes to xidb
    // (MFMA operand, zeros kept) and its strictly-upper part into the diagonal sub-block.
    if (is_diag) {
        if (wave == 0) {
#pragma unroll
            for (int c = 0; c < SB; ++c)
                if (c <= lane) Mblk[(long long)row * ld + c0 + c] = a[c];
        }
    } else if (valid) {
        if (is_newid) {
            const int q = idrow - c0;
#pragma unroll
            for (int c = 0; c < SB; c += 2) *(d2*)&xidb[q][c] = (d2){a[c], a[c + 1]};
#pragma unroll
            for (int c = 0; c < SB; ++c)
                if (c > q) Ls[row][c0 + c] = a[c];
        } else {
#pragma unroll
            for (int c = 0; c < SB; c += 2) *(d2*)&Ls[row][c0 + c] = (d2){a[c], a[c + 1]};
        }
    }
}

// Rank-16 update by block column jb of the tiles right of it:
//   lower tiles  C(bi,bk) -= X(bi).X(bk)^T    (blocks below the diagonal one, bk <= bi)
//   upper tiles  U(ib,bk) -= Xid(ib).X(bk)^T  (identity rows ib <= jb; ib == jb reads xidb)
// `priority`: only the tiles of the next block column (bk == 0), one per wave `w` of `nw`;
// otherwise all the other tiles (bk >= 1), two independent tiles in flight per wave.
__device__ __forceinline__ void update_tiles(double (*Ls)[LS], const double (*xidb)[XS], int jb, bool priority,
                                             int w, int nw, int fr, int fq) {
    const int c0 = jb * SB, base = c0 + SB;
    const int nb16 = (NB - base) / SB;
    // tile list of this phase, indexed 0..count-1
    //   priority: [0, nb16) lower (bi = t, bk = 0); [nb16, nb16 + jb + 1) upper (ib = t - nb16, bk = 0)
    //   rest:     lower tiles with bk >= 1: nb16*(nb16-1)/2; upper: (jb+1)*(nb16-1)
    const int nlow = priority ? nb16 : nb16 * (nb16 - 1) / 2;
    const int count = priority ? nb16 + jb + 1 : nlow + (jb + 1) * (nb16 - 1);
    auto decode = [&](int t, int& crow, int& ccol, const double*& ap, int& astride) {
        if (t < nlow) {
            int bi, bk;
            if (priority) { bi = t; bk = 0; }
            else {          // pairs 1 <= bk <= bi < nb16, row-major over bi
                bi = 1;
                while (bi * (bi + 1) / 2 <= t) ++bi;
                bk = t - bi * (bi - 1) / 2 + 1;
            }
            crow = base + SB * bi; ccol = base + SB * bk;
            ap = &Ls[crow][c0]; astride = LS;
        } else {
            const int u = t - nlow;
            int ib, bk;
            if (priority) { ib = u; bk = 0; }
            else { ib = u / (nb16 - 1); bk = u - ib * (nb16 - 1) + 1; }
            crow = SB * ib; ccol = base + SB * bk;
            if (ib == jb) { ap = &xidb[0][0]; astride = XS; }
            else          { ap = &Ls[crow][c0]; astride = LS; }
        }
    };
    for (int t0 = w; t0 < count; t0 += 2 * nw) {
        const int t1 = t0 + nw;
        const bool has1 = t1 < count;
        int r0_, q0_, s0_, r1_ = 0, q1_ = 0, s1_ = LS;
        const double *ap0, *ap1 = &Ls[0][0];
        decode(t0, r0_, q0_, ap0, s0_);
        if (has1) decode(t1, r1_, q1_, ap1, s1_);
        d4 ca, cb = (d4){0.0, 0.0, 0.0, 0.0};
        double fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ca[r] = Ls[r0_ + fq + 4 * r][q0_ + fr];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            fa0[u] = -ap0[fr * s0_ + 4 * u + fq];
            fb0[u] = Ls[q0_ + fr][c0 + 4 * u + fq];
        }
        if (has1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cb[r] = Ls[r1_ + fq + 4 * r][q1_ + fr];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                fa1[u] = -ap1[fr * s1_ + 4 * u + fq];
                fb1[u] = Ls[q1_ + fr][c0 + 4 * u + fq];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { fa1[u] = 0.0; fb1[u] = 0.0; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ca = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[u], fb0[u], ca, 0, 0, 0);
            if (has1) cb = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[u], fb1[u], cb, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Ls[r0_ + fq + 4 * r][q0_ + fr] = ca[r];
        if (has1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Ls[r1_ + fq + 4 * r][q1_ + fr] = cb[r];
        }
    }
}

// Mblk: top-left of the 128x128 diagonal block (row-major, ld).  Linv / LinvT: where inv(L_kk) and
// its transpose go (row-major, ldinv): the diagonal 128-blocks of the super-block inverses.
//
// ONE elimination produces both the factor and its inverse.  Right-looking Cholesky applied to the
// augmented matrix [A; I] leaves [L; L^-T]: the extra identity rows are just more panel rows of the
// triangular solve X.L^T = B with B = I.  Row i of L^-T is zero left of column i, so its entries live
// in the (otherwise unused) strict upper triangle of the LDS image; the diagonal 1/L_ii goes to dinv.
//
// 8 block columns of 16.  Each is eliminated by waves 0..2 on 144 rows held one per lane: lanes 0-15
// of every wave replicate the 16 diagonal rows (no wave waits on another), lanes 16-63 carry the
// 128 other rows (panel rows below, identity rows above / inside the block).  The rank-16 update of
// everything to the right runs as 16x16 MFMA tiles and is software-pipelined against the
// elimination: first the 8 tiles of the NEXT block column (one per wave), barrier, then waves 0..2
// eliminate that column while waves 3..15 apply the rest of the update.
__global__ __launch_bounds__(DT) void potrf_diag_kernel(double* __restrict__ Mblk, long long ld,
                                                        double* __restrict__ Linv, double* __restrict__ LinvT,
                                                        long long ldinv, int32_t* info, int global_row0,
                                                        long long* stamps, BatchK bk) {
    if (batch_done(bk)) return;
    Mblk = batch_ptr(Mblk, bk); Linv = batch_ptr(Linv, bk); LinvT = batch_ptr(LinvT, bk); info = batch_ptr(info, bk);
#define STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[i] = clock64(); } while (0)
    __shared__ __attribute__((aligned(16))) double Ls[NB][LS];            // 133,120 B
    __shared__ __attribute__((aligned(16))) double xid[2][SB][XS];        // eliminated NEW identity rows (double-buffered)
    __shared__ __attribute__((aligned(16))) double scr[3][2][SB];         // per-wave column scratch
    __shared__ __attribute__((aligned(16))) double dinv[NB];              // 1 / L_ii
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int NW = DT / 64;

    for (int e = tid * 2; e < NB * NB; e += 2 * DT) {
        const int r = e >> 7, c = e & 127;
        d2 v = *(const d2*)(Mblk + (long long)r * ld + c);
        if (c > r) v[0] = 0.0;
        if (c + 1 > r) v[1] = 0.0;
        *(d2*)&Ls[r][c] = v;
    }
    lds_barrier();
    STAMP(0);
    if (wave < 3) eliminate_block_column(Ls, xid[0], scr[wave], dinv, 0, wave, lane, info, global_row0, Mblk, ld);
    lds_barrier();
    STAMP(1);
    // Everything that becomes final with block column jb -- columns c0..c0+15 of L (below the diagonal
    // sub-block, which the eliminating wave stored itself), of L^-T (rows 0..c0+15) and rows
    // c0..c0+15 of L^-1 -- is written out by threads [t0, t0+nt) as fire-and-forget global stores.
    // Only the triangles that can be non-zero are stored: the other halves of the inverse slabs are
    // zero from allocation on and nothing ever writes them.
    auto store_block_column = [&](int jb, int t, int nt) {
        const int c0 = jb * SB;
        for (int e = t; e < (NB - c0 - SB) * (SB / 2); e += nt) {
            const int r = c0 + SB + e / (SB / 2), c = c0 + 2 * (e % (SB / 2));
            *(d2*)(Mblk + (long long)r * ld + c) = *(const d2*)&Ls[r][c];
        }
        for (int e = t; e < (c0 + SB) * (SB / 2); e += nt) {       // LinvT[i][c0+c] = U[i][c0+c], c0+c >= i
            const int i = e / (SB / 2), c = c0 + 2 * (e % (SB / 2));
            if (c + 1 >= i) {
                d2 w;
                w[0] = (c == i) ? dinv[i] : ((c > i) ? Ls[i][c] : 0.0);
                w[1] = (c + 1 == i) ? dinv[i] : Ls[i][c + 1];
                *(d2*)(LinvT + (long long)i * ldinv + c) = w;
            }
        }
        for (int e = t; e < SB * ((c0 + SB) / 2); e += nt) {       // Linv[r][i] = U[i][r], i <= r
            const int r = c0 + e / ((c0 + SB) / 2), i = 2 * (e % ((c0 + SB) / 2));
            if (i <= r) {
                d2 v;
                v[0] = (i == r) ? dinv[r] : Ls[i][r];
                v[1] = (i + 1 == r) ? dinv[r] : ((i + 1 < r) ? Ls[i + 1][r] : 0.0);
                *(d2*)(Linv + (long long)r * ldinv + i) = v;
            }
        }
    };
    for (int jb = 0; jb < NSB - 1; ++jb) {
        update_tiles(Ls, xid[jb & 1], jb, true, wave, NW, fr, fq);          // the next block column first
        lds_barrier();
        if (jb == 0) STAMP(2);
        if (wave < 3) {
            __builtin_amdgcn_s_setprio(3);       // the serial chain outranks the throughput work sharing its SIMDs
            eliminate_block_column(Ls, xid[(jb + 1) & 1], scr[wave], dinv, jb + 1, wave, lane, info, global_row0, Mblk, ld);
            __builtin_amdgcn_s_setprio(0);
        } else {
            update_tiles(Ls, xid[jb & 1], jb, false, wave - 3, NW - 3, fr, fq);
            store_block_column(jb, tid - 192, DT - 192);
        }
        lds_barrier();
        if (jb == 0) STAMP(3);
    }
    STAMP(6);
    store_block_column(NSB - 1, tid, DT);
    STAMP(7);
#undef STAMP
}

long long* g_diag_stamps = nullptr;  // debug: device buffer of 8 cycle stamps for block 0 (scripts/)
// Two-level right-looking factorisation.  Inside an outer panel of OUTER 128-blocks the steps are
//   diag(j) -> panel solve(j) -> update of the REST OF THE OUTER PANEL only (K = 128, few tiles)
// and the big trailing update runs once per outer panel with K = OUTER*128: 4x fewer read-modify-write
// passes over the trailing matrix (at K = 128 that update is bound by the C-tile traffic, 16 flop/B).
constexpr int OUTER = POTRF_OUTER;

hipError_t potrf_clear_info(int32_t* info, hipStream_t st, const Batch& bt) {
    // (a finished LP of a batch has its info word cleared too: its status record already holds the value)
    return bt.count == 1 ? hipMemsetAsync(info, 0, sizeof(int32_t), st)
                         : hipMemset2DAsync(info, (size_t)bt.stride, 0, sizeof(int32_t), (size_t)bt.count, st);
}

// The dependent chain of one outer panel [J0, J1) of 128-blocks: diag(j) -> panel solve(j) (all rows below) -> update
// of the rest of the outer panel's columns (K = 128, few tiles).
hipError_t potrf_panel_chain(double* M, int64_t ld, int mp, const FactorPlan& plan, int32_t* info, hipStream_t st,
                             const Batch& bt, int J0, int J1) {
    hipError_t e;
    const int nb = mp / NB;
    for (int j = J0; j < J1; ++j) {
        const int64_t o = (int64_t)j * NB;
        double* diag = M + o * ld + o;
        double* linv = plan.blk_inv(j);
        const int ldinv = plan.blk_ld(j);
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, 1, bt.count), dim3(DT), 0, st, diag, (long long)ld, linv,
                           plan.blk_invT(j), (long long)ldinv, info, (int)o,
                           (long long*)(j == 0 ? g_diag_stamps : nullptr), batch_k(bt));
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int rem = nb - j - 1;
        if (rem <= 0) break;
        double* panel = M + (o + NB) * ld + o;   // rows below block j, column block j
        GemmArgs t{};
        t.P = panel; t.ldp = ld; t.Q = linv; t.ldq = ldinv; t.s = nullptr;
        t.C = panel; t.ldc = ld; t.K = NB; t.alpha = 1.0; t.beta = 0.0;
        // few tiles -> latency-bound: 32-row x 128-col tiles put 4x as many CUs on the panel, and a
        // workgroup still owns whole rows, so the product may overwrite its own input
        t.tile_edge = 32;
        t.ntiles = 4 * rem; t.tiles_lower = 0; t.ntj = 1; t.tile_list = nullptr;
        t.diag_pad_from = -1; t.ws = nullptr; t.nwg = t.ntiles; t.batch = bt;
        e = launch_gemm_nt(t, st);
        if (e != hipSuccess) return e;
        const int ncols = J1 - j - 1;            // column blocks of the outer panel right of j
        if (ncols > 0) {
            // rows j+1..nb x columns j+1..J1-1 -= L[rows, j] . L[cols, j]^T  (rectangular grid of
            // 64x64 tiles; the few tiles above the diagonal are computed too and never read)
            GemmArgs c{};
            c.P = panel; c.ldp = ld; c.Q = panel; c.ldq = ld; c.s = nullptr;
            c.C = M + (o + NB) * ld + (o + NB); c.ldc = ld; c.K = NB; c.alpha = -1.0; c.beta = 1.0;
            c.tile_edge = 64; c.tiles_lower = 0; c.ntj = 2 * ncols; c.ntiles = (2 * rem) * (2 * ncols);
            c.tile_list = nullptr; c.diag_pad_from = -1; c.ws = nullptr; c.nwg = c.ntiles; c.batch = bt;
            e = launch_gemm_nt(c, st);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

// Right-looking trailing update behind the outer panel [J0, J1): A22 -= L21 . L21^T with K = (J1 - J0) * 128.
hipError_t potrf_trailing_update(double* M, int64_t ld, int mp, hipStream_t st, const Batch& bt, int J0, int J1) {
    const int nb = mp / NB;
    const int remT = nb - J1;                    // trailing blocks beyond the outer panel
    if (remT <= 0) return hipSuccess;
    const int64_t oJ0 = (int64_t)J0 * NB, oJ1 = (int64_t)J1 * NB;
    GemmArgs u{};
    u.P = M + oJ1 * ld + oJ0; u.ldp = ld; u.Q = u.P; u.ldq = ld; u.s = nullptr;
    u.C = M + oJ1 * ld + oJ1; u.ldc = ld; u.K = (J1 - J0) * NB; u.alpha = -1.0; u.beta = 1.0;
    u.tiles_lower = 1; u.ntj = 0; u.tile_list = nullptr;
    // 64x64 tiles (4 workgroups per CU) until the 128x128 ones would fill the chip's 512 slots about twice:
    // below that a launch lasts one K = 512 tile (~150 us at 128, ~60 at 64) whatever its tile count
    // (factorisation at m = 4096: 2400 -> 2311 us)
    if (remT * (remT + 1) / 2 < 1024) { u.tile_edge = 64; u.ntiles = (2 * remT) * (2 * remT + 1) / 2; }
    else                             { u.tile_edge = 128; u.ntiles = remT * (remT + 1) / 2; }
    u.diag_pad_from = -1; u.ws = nullptr; u.nwg = u.ntiles; u.batch = bt;
    return launch_gemm_nt(u, st);
}

// inverses of the diagonal super-blocks from the 128-block inverses: doubling levels, each a
// grouped launch of stage A (T^T = Inv11^T.L21^T) then stage B (Inv21 = -Inv22.T and its transpose)
hipError_t potrf_superblock_inverses(const FactorPlan& plan, hipStream_t st, const Batch& bt) {
    for (const auto& stg : plan.stages) {
        hipError_t e = launch_gemm_grouped(plan.descs_dev + stg.first, stg.second, st, bt, plan.merge_edge);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_potrf(double* M, int64_t ld, int mp, const FactorPlan& plan, int32_t* info, hipStream_t st,
                        const Batch& bt) {
    hipError_t e = potrf_clear_info(info, st, bt);
    if (e != hipSuccess) return e;
    const int nb = mp / NB;
    for (int J0 = 0; J0 < nb; J0 += OUTER) {
        const int J1 = J0 + OUTER < nb ? J0 + OUTER : nb;
        if ((e = potrf_panel_chain(M, ld, mp, plan, info, st, bt, J0, J1)) != hipSuccess) return e;
        if ((e = potrf_trailing_update(M, ld, mp, st, bt, J0, J1)) != hipSuccess) return e;
    }
    return potrf_superblock_inverses(plan, st, bt);
}

}  // namespace lpipm
