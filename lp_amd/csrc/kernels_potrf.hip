// kernels_potrf.hip -- blocked lower Cholesky M = L.L^T on device (replaces `M.cholesky()`,
// newton_equations.rs:129-131; the reference's default backend is an unblocked scalar loop).
//
// Right-looking, block size NB = 128 (== the MFMA GEMM tile):
//   for each block column k:
//     1. potrf_diag_kernel  (ONE workgroup): factor the 128x128 diagonal block in LDS and form
//        inv(L_kk); both are needed downstream (inv(L_kk) turns the panel TRSM into a GEMM and the
//        triangular solves into block mat-vecs).  See the comment in front of the kernel.
//     2. L21 = A21 . inv(L_kk)^T           -> MFMA NT-GEMM, in place
//     3. A22 -= L21 . L21^T (lower tiles)  -> MFMA NT-GEMM, alpha=-1, beta=1
// A non-positive pivot is recorded in `info` (1 + its global index, first one wins) and the
// factorisation continues on NaNs; the host maps info != 0 to NumericalProblem exactly where the
// reference maps a failed `cholesky()` (newton_equations.rs:59-63).
#include "lpipm_internal.hpp"
#include <utility>
#include <type_traits>

namespace lpipm {

constexpr int SB  = 16;        // sub-block edge inside the diagonal block
constexpr int NSB = NB / SB;   // 8
constexpr int LS  = NB + 2;    // LDS row stride in doubles (260 dwords: rows 4 banks apart)
constexpr int DT  = 1024;      // threads of the diagonal-block kernel: 16 waves, 4 per SIMD

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

// Barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the fire-and-forget
// global stores of finished columns (thousands of cycles per block column).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------
// potrf_diag_kernel: ONE workgroup factors a 128x128 diagonal block AND inverts the factor, by eliminating the augmented
// matrix [A; I] in LDS: right-looking Cholesky applied to [A; I] leaves [L; L^-T] (the identity rows are just more panel
// rows of the triangular solve X.L^T = B with B = I).  8 block columns of 16.
//
// What the hardware gives (probes under scripts/diag, MI355X, shader-clock cycles):
//   * a wave issues at most one instruction every ~4-5 cycles whatever its kind; a dependent v_fma_f64 returns after 8,
//     v_rcp_f64 after 20: the chain of one pivot (reciprocal seed, two Newton steps, multiplier, next pivot) is ~80;
//   * FP64 MFMA is no faster than FP64 VALU (v_mfma_f64_16x16x4: one per ~75 cycles and SIMD = 1024 FMA, the rate of
//     v_fma_f64) and runs on the same pipe: a wave streaming v_fma_f64 and a wave issuing MFMAs on ONE SIMD halve each other;
//   * an LDS round trip (write, read) is ~110 cycles, a wave on its own moves ~40 B/clk into LDS;
//   * v_fmac_f64_dpp / v_mov_b64_dpp with row_newbcast:k read lane k of the lane's own row of 16 at full rate
//     (v_rcp_f64_dpp assembles but reads zero).
// Round 1's kernel (three waves eliminating 48 rows each, the multiplier column broadcast through LDS, the 1/sqrt scaling
// and three divergent write-back paths on the chain, tile decode on the VALU because the wave index was a vector value)
// took 33 us = ~66k cycles; its stamps showed 4400 cycles per 16 pivots where the chain needs 1300, and the tile updates
// of the waves sharing a SIMD with an eliminating wave only ran once that wave was done.  This one: ~50k cycles.
//   * E(k), elimination of block column k: ONE wave, at priority 3.  Lane = 16 rho + l holds a replica of diagonal row l
//     and two of the 128 other rows (panel rows below, the identity rows of earlier block columns, the 16 entering now);
//     the multiplier column is broadcast inside the FMA (v_fmac_f64_dpp), so the pivot loop touches no memory and no SGPR
//     and is written as an explicit schedule (pivot_step).  Three SIMDs stay free of it.
//   * Columns stay UNSCALED in LDS (an LDL^T elimination); the rank-16 updates scale both operands by 1/sqrt(pivot) -- the
//     products the stored factor consists of -- and the global stores scale on the way out: nothing but the pivots is on
//     the chain.  (Scaling only one operand by 1/pivot is the same algebra and cost two C4 members their last digit.)
//   * P(k), the 8 tiles of block column k+1 (waves 0..7, counted in an LDS word the eliminating wave waits for), R(k), the
//     tiles right of it (MFMA workers: two waves per free SIMD, operands of the next tile loaded behind the first MFMA of
//     the current one, tile addresses from a table built at kernel start), and the stores of everything final with k
//     (store workers) all run beside E(k+1): one workgroup barrier per block column.
constexpr int XS3 = SB + 2;    // row stride of the new-identity-row operand block (conflict-free MFMA reads, 16 B aligned)

template <int K> __device__ __forceinline__ void fmac_bcast(double& acc, const double& src, const double& m) {
    // acc += (src of lane K of this lane's row of 16) * m
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(K));
}
template <int K> __device__ __forceinline__ double mov_bcast(const double& src) {
    double r;
    // s_nop 1: a DPP read needs two wait states after the VALU write of its source (the pivot entry was just updated).
    // (v_rcp_f64_dpp assembles but reads its source as zero on gfx950 -- scripts/diag/dpp_step_probe.cpp: broadcast first.)
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(K));
    return r;
}

// One pivot step of the elimination, J = pivot column.  a = this row-of-16's replica of the diagonal rows (lane l: row l),
// o[n] = the lane's other rows.  tn* = -(multipliers of step J-1).  Order of the DPP ops (asm volatile) and, through
// sched_barrier, of everything else IS the schedule:
//   reciprocal seed and broadcast of pivot J; in their shadow the bulk updates of step J-1 (sources: column J-1, final
//   since the step before); the Newton steps with a filler in each bubble; the multipliers; the next pivot column's
//   entry of the replica.  A DPP read needs two instructions between it and the write of its source: the fillers that
//   follow the last write of a[J] provide them.
template <int J, int NARR>
__device__ __forceinline__ void pivot_step(double (&a)[SB], double (&o)[NARR][SB], double& tna, double (&tno)[NARR],
                                           double* const (&op)[NARR], double* da, bool write_a) {
#define SB0 __builtin_amdgcn_sched_barrier(0)
    const double pv = mov_bcast<J>(a[J]);
    SB0;
    double r = __builtin_amdgcn_rcp(pv);
    SB0;
    // bulk of step J-1 except the last four ops, which go into the Newton bubbles
    constexpr int NA = (J == 0) ? 0 : SB - 1 - J;            // a[k], k = J+1..15 (a[J] was updated on the chain)
    constexpr int NO = (J == 0) ? 0 : SB - J;                // o[n][k], k = J..15
    auto filler = [&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (J > 0) {
            if constexpr (i < NO * NARR) {                   // other rows first: o[n][J] feeds this step's multiplier
                constexpr int n = i % NARR, k = J + i / NARR;
                fmac_bcast<k>(o[n][k], a[J - 1], tno[n]);
            } else {
                constexpr int k = J + 1 + (i - NO * NARR);
                fmac_bcast<k>(a[k], a[J - 1], tna);
            }
        }
    };
    constexpr int NF = NO * NARR + NA;
    constexpr int TAIL = NF > 4 + NARR ? 4 : 0;              // keep the o[n][J] updates out of the bubbles
    [&]<int... I>(std::integer_sequence<int, I...>) { (filler(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, NF - TAIL>{});
    SB0;
    // columns J-1 and J are final now: write them back here, one 16-byte store per row array in the shadow of the chain,
    // instead of 8 per array after the loop (a wave on its own gets ~40 B/clk out of the LDS store path: the 24 stores of
    // the single-wave variant took 1150 cycles)
    if constexpr ((J & 1) == 1) {
#pragma unroll
        for (int n = 0; n < NARR; ++n) *(d2*)(op[n] + J - 1) = (d2){o[n][J - 1], o[n][J]};
        if (write_a) *(d2*)(da + J - 1) = (d2){a[J - 1], a[J]};
        SB0;
    }
    double e = fma(-pv, r, 1.0); SB0;
    if constexpr (TAIL > 0) filler(std::integral_constant<int, NF - 4>{});
    SB0; r = fma(r, e, r); SB0;
    if constexpr (TAIL > 0) filler(std::integral_constant<int, NF - 3>{});
    SB0; e = fma(-pv, r, 1.0); SB0;
    if constexpr (TAIL > 0) filler(std::integral_constant<int, NF - 2>{});
    SB0; r = fma(r, e, r); SB0;
    if constexpr (TAIL > 0) filler(std::integral_constant<int, NF - 1>{});
    SB0;
    tna = -a[J] * r; SB0;
#pragma unroll
    for (int n = 0; n < NARR; ++n) tno[n] = -o[n][J] * r;
    SB0;
    if constexpr (J + 1 < SB) fmac_bcast<J + 1>(a[J + 1], a[J], tna);
    SB0;
#undef SB0
}

// Eliminates block column jb on all 144 rows.  Wave ew of 2/NARR eliminating waves; lane = 16 rho + l holds the replica of
// diagonal row l and NARR other rows: 16-row blocks beta = (ew NARR + n) 4 + rho, in the order panel blocks jb+1..7,
// earlier identity blocks 0..jb-1, the identity block entering with jb (starts as the identity, ends in xidb).
// Columns are left unscaled: row -= (row[j] / pivot_j) . S(., j).
template <int NARR>
__device__ __forceinline__ void eliminate16(double (*Ls)[LS], double (*xidb)[XS3], double (*doutb)[SB],
                                            const double (*eye)[SB], int jb, int ew, int lane) {
    const int c0 = jb * SB;
    const int l = lane & (SB - 1), rho = lane >> 4;
    double a[SB], o[NARR][SB];
    double* op[NARR];
    {
        const double* src = &Ls[c0 + l][c0];
#pragma unroll
        for (int c = 0; c < SB; c += 2) { const d2 x = *(const d2*)(src + c); a[c] = x[0]; a[c + 1] = x[1]; }
    }
#pragma unroll
    for (int n = 0; n < NARR; ++n) {
        const int beta = (ew * NARR + n) * 4 + rho;
        const int npb = NSB - 1 - jb;                        // panel blocks
        const double* src;
        if (beta < npb)          { op[n] = &Ls[(jb + 1 + beta) * SB + l][c0]; src = op[n]; }
        else if (beta < NSB - 1) { op[n] = &Ls[(beta - npb) * SB + l][c0];    src = op[n]; }
        else                     { op[n] = &xidb[l][0];                       src = &eye[l][0]; }
#pragma unroll
        for (int c = 0; c < SB; c += 2) { const d2 x = *(const d2*)(src + c); o[n][c] = x[0]; o[n][c + 1] = x[1]; }
    }
    double tna = 0.0, tno[NARR];
#pragma unroll
    for (int n = 0; n < NARR; ++n) tno[n] = 0.0;
    const bool write_a = ew == 0 && rho == 0;
    double* const da = &doutb[l][0];
    [&]<int... J>(std::integer_sequence<int, J...>) { (pivot_step<J, NARR>(a, o, tna, tno, op, da, write_a), ...); }(std::make_integer_sequence<int, SB - 1>{});
    // the bulk of the last step that ran (J = 14): column 15 of the other rows (a[15] was updated on the chain)
#pragma unroll
    for (int n = 0; n < NARR; ++n) fmac_bcast<SB - 1>(o[n][SB - 1], a[SB - 2], tno[n]);
#pragma unroll
    for (int n = 0; n < NARR; ++n) *(d2*)(op[n] + SB - 2) = (d2){o[n][SB - 2], o[n][SB - 1]};
    if (write_a) *(d2*)(da + SB - 2) = (d2){a[SB - 2], a[SB - 1]};
}

// Wave-uniform tile address of the rank-16 update by block column jb.  PRIO numbering (block column jb+1 only):
// t in [0, nb16): lower tile (bi = t, bk = 0), then jb+1 upper tiles; otherwise: the nb16(nb16-1)/2 lower tiles with
// 1 <= bk <= bi < nb16 (row-major over bi), then (jb+1)(nb16-1) upper tiles.  ap/astride: the tile's row operand X(bi)
// (the identity rows of block jb itself live in xidb).
template <bool PRIO>
__device__ __forceinline__ void tile_decode(const double (*Ls)[LS], const double (*xidb)[XS3], int jb, int t, int& crow,
                                            int& ccol, const double*& ap, int& astride) {
    const int c0 = jb * SB, base = c0 + SB;
    const int nb16 = (NB - base) / SB;
    const int nlow = PRIO ? nb16 : nb16 * (nb16 - 1) / 2;
    int bk = 0;
    if (t < nlow) {
        int bi = t;
        if (!PRIO) {                   // t = bi (bi - 1) / 2 + bk - 1; compares instead of a loop: a taken scalar branch costs ~20 cycles
            bi = 1 + (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15);
            bk = t - bi * (bi - 1) / 2 + 1;
        }
        crow = base + SB * bi;
        ap = &Ls[crow][c0]; astride = LS;
    } else {
        int u = t - nlow, ib = u;
        if (!PRIO) {
            const int n1 = nb16 - 1;
            ib = (u >= n1) + (u >= 2 * n1) + (u >= 3 * n1) + (u >= 4 * n1) + (u >= 5 * n1) + (u >= 6 * n1);
            bk = u - ib * n1 + 1;
        }
        crow = SB * ib;
        if (ib == jb) { ap = &xidb[0][0]; astride = XS3; }
        else          { ap = &Ls[crow][c0]; astride = LS; }
    }
    ccol = base + SB * bk;
}

// One or two 16x16 tiles of the rank-16 update:  C -= (X(bi) . diag(1/pivot)) . X(bk)^T  with the unscaled columns X.
// Every LDS access of a lane is 4 contiguous doubles (two ds_read_b128 / ds_write_b128 per fragment; one wave alone gets a
// fraction of the LDS rate on 8-byte accesses and the 24 reads of a tile pair took ~600 cycles):
//   * contraction index: MFMA u of 4 takes k = 4 fq + u from lane group fq (any split of the 16 k over the four
//     instructions is the same sum), so a lane's operand registers are X[row][4fq .. 4fq+3];
//   * the MFMA computes the TRANSPOSED tile (operands swapped) with the rows of its first operand permuted: lane fr of that
//     operand holds row p(fr) = 4 (fr & 3) + (fr >> 2) of X(bk), so accumulator register r of lane (fr, fq) -- hardware row
//     fq + 4r -- is C[fr][4fq + r].
// pv[u]: 1/pivot of column 4fq + u, or (RAWPIV) the pivot itself, inverted here once every load is in flight.
struct TileAddr { int crow, ccol, astride; const double* ap; };
struct TileRegs { d4 c, fa, fb; };
__device__ __forceinline__ d4 lds_ld4(const double* p) {
    const d2 lo = *(const d2*)p, hi = *(const d2*)(p + 2);
    return (d4){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void lds_st4(double* p, const d4& v) { *(d2*)p = (d2){v[0], v[1]}; *(d2*)(p + 2) = (d2){v[2], v[3]}; }
__device__ __forceinline__ void tile_load(const double (*Ls)[LS], const TileAddr& t, int c0, int fr, int fq, TileRegs& g) {
    const int pr = 4 * (fr & 3) + (fr >> 2);
    g.c = lds_ld4(&Ls[t.crow + fr][t.ccol + 4 * fq]);
    g.fa = lds_ld4(t.ap + fr * t.astride + 4 * fq);          // X(bi)[fr][4fq..]
    g.fb = lds_ld4(&Ls[t.ccol + pr][c0 + 4 * fq]);           // X(bk)[p(fr)][4fq..]
}
template <bool TWO>
__device__ __forceinline__ void tile_mfma_store(double (*Ls)[LS], const TileAddr& t0, const TileAddr& t1, int fr, int fq,
                                                const double pv[4], TileRegs& g0, TileRegs& g1) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        g0.c = __builtin_amdgcn_mfma_f64_16x16x4f64(g0.fb[u] * pv[u], -(g0.fa[u] * pv[u]), g0.c, 0, 0, 0);
        if (TWO) g1.c = __builtin_amdgcn_mfma_f64_16x16x4f64(g1.fb[u] * pv[u], -(g1.fa[u] * pv[u]), g1.c, 0, 0, 0);
    }
    lds_st4(&Ls[t0.crow + fr][t0.ccol + 4 * fq], g0.c);
    if (TWO) lds_st4(&Ls[t1.crow + fr][t1.ccol + 4 * fq], g1.c);
}
// The same loads issued as inline asm, which the compiler's s_waitcnt insertion does not track: around a loop it only ever
// emitted lgkmcnt(0) in front of the MFMAs, i.e. it waited for the tile prefetched a moment ago as well.  tile_wait<N>
// leaves the N newest LDS operations in flight (they return in order) and ties the registers to the wait.
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
struct TileRegs2 { d2 c[2], fa[2], fb[2]; };
__device__ __forceinline__ void tile_load_async(const double (*Ls)[LS], const TileAddr& t, int c0, int fr, int fq, TileRegs2& g) {
    const int pr = 4 * (fr & 3) + (fr >> 2);
    const unsigned ac = lds_off(&Ls[t.crow + fr][t.ccol + 4 * fq]), aa = lds_off(t.ap + fr * t.astride + 4 * fq),
                   ab = lds_off(&Ls[t.ccol + pr][c0 + 4 * fq]);
    asm volatile("ds_read_b128 %0, %1" : "=v"(g.c[0]) : "v"(ac));
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(g.c[1]) : "v"(ac));
    asm volatile("ds_read_b128 %0, %1" : "=v"(g.fa[0]) : "v"(aa));
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(g.fa[1]) : "v"(aa));
    asm volatile("ds_read_b128 %0, %1" : "=v"(g.fb[0]) : "v"(ab));
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(g.fb[1]) : "v"(ab));
}
template <int N> __device__ __forceinline__ void tile_wait(TileRegs2& g) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(g.c[0]), "+v"(g.c[1]), "+v"(g.fa[0]), "+v"(g.fa[1]), "+v"(g.fb[0]), "+v"(g.fb[1]) : "n"(N));
}
__device__ __forceinline__ void tile_mfma_store2(double (*Ls)[LS], const TileAddr& t, int fr, int fq, const double pv[4], TileRegs2& g) {
    d4 c = (d4){g.c[0][0], g.c[0][1], g.c[1][0], g.c[1][1]};
#pragma unroll
    for (int u = 0; u < 4; ++u) c = __builtin_amdgcn_mfma_f64_16x16x4f64(g.fb[u >> 1][u & 1] * pv[u], -(g.fa[u >> 1][u & 1] * pv[u]), c, 0, 0, 0);
    lds_st4(&Ls[t.crow + fr][t.ccol + 4 * fq], c);
}
// One tile with the pivots taken raw from the staging block and inverted once every load is in flight (the P phase).
__device__ __forceinline__ void update_tile_rawpiv(double (*Ls)[LS], const TileAddr& t, int c0, double pv[4], int fr, int fq) {
    TileRegs g;
    tile_load(Ls, t, c0, fr, fq, g);
    __builtin_amdgcn_sched_barrier(0);
    double h[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { h[u] = 0.5 * pv[u]; pv[u] = __builtin_amdgcn_rsq(pv[u]); }
#pragma unroll
    for (int it = 0; it < 2; ++it) {                       // rsqrt_nr, the four chains interleaved
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = fma(-h[u] * pv[u], pv[u], 0.5);
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = fma(pv[u], w[u], pv[u]);
    }
    tile_mfma_store<false>(Ls, t, t, fr, fq, pv, g, g);
}

// Mblk: top-left of the 128x128 diagonal block (row-major, ld).  Linv / LinvT: where inv(L_kk) and its transpose go
// (row-major, ldinv): the diagonal 128-blocks of the super-block inverses.  The elimination of [A; I] leaves [L; L^-T]:
// row i of L^-T lives in the strict upper triangle of the LDS image right of its own 16-block and, inside its own
// block, in xid.  The strict upper triangle of the diagonal 16x16 sub-blocks of Mblk is written as zeros (nothing
// reads it: this kernel itself ignores it on load).
template <int NARR>
__global__ __launch_bounds__(DT) void potrf_diag_kernel(double* __restrict__ Mblk, long long ld,
                                                         double* __restrict__ Linv, double* __restrict__ LinvT,
                                                         long long ldinv, int32_t* info, int global_row0,
                                                         long long* stamps, BatchK bk) {
    if (batch_done(bk)) return;
    Mblk = batch_ptr(Mblk, bk); Linv = batch_ptr(Linv, bk); LinvT = batch_ptr(LinvT, bk); info = batch_ptr(info, bk);
#define STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[i] = clock64(); } while (0)
#define WSTAMP(i) do { if (stamps && lane == 0) stamps[(i) + wave] = clock64(); } while (0)
    constexpr int NEW = 2 / NARR;                                         // eliminating waves: 0 .. NEW-1
    __shared__ __attribute__((aligned(16))) double Ls[NB][LS];            // 133,120 B
    __shared__ __attribute__((aligned(16))) double xid[2][SB][XS3];       // identity rows of block column k (double-buffered)
    __shared__ __attribute__((aligned(16))) double dout[2][SB][SB];       // eliminated diagonal rows (double-buffered)
    __shared__ __attribute__((aligned(16))) double eye[SB][SB];
    __shared__ __attribute__((aligned(16))) double dinv[NB];              // 1 / L_ii = 1/sqrt(pivot)
    __shared__ int wsimd[DT / 64];                                        // SIMD of every wave
    __shared__ int wrole[DT / 64];                                        // role beside the elimination (below)
    __shared__ int nrole[2];                                              // MFMA workers, store workers
    __shared__ int pdone[NSB], ready[NSB];                                // P tiles done / scales and tile table ready, per block column
    __shared__ __attribute__((aligned(16))) int ttab[NSB - 1][32][4];     // R tiles of every block column: LDS byte offsets
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave-uniform: tile decode and branches go scalar
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int NW = DT / 64;
    int my_simd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(my_simd));

    // The image: 8 pairs per thread, all requested before anything else -- a loop of load / mask / LDS store paid one global
    // round trip per pair (6700 cycles of prologue, 9000 with the role set-up behind it) -- and stored once the roles below
    // have been worked out in their shadow (lds_barrier does not wait for vmcnt): 5000.
    d2 img[NB * NB / (2 * DT)];
#pragma unroll
    for (int i = 0; i < NB * NB / (2 * DT); ++i) {
        const int e = tid * 2 + i * 2 * DT;
        img[i] = *(const d2*)(Mblk + (long long)(e >> 7) * ld + (e & 127));
    }
    if (tid < SB * SB) eye[tid >> 4][tid & 15] = ((tid >> 4) == (tid & 15)) ? 1.0 : 0.0;
    if (lane == 0) wsimd[wave] = my_simd;
    if (tid < NSB) { pdone[tid] = 0; ready[tid] = 0; }
    lds_barrier();
    // Roles beside the elimination.  FP64 MFMA and v_fma_f64 share a SIMD's double-precision pipe and the eliminating wave
    // runs at priority 3, so the waves on its SIMD stay out of the way.  On every other SIMD the first two waves stream the
    // MFMA tiles (with four waves in step on one SIMD the chains only took turns), the others do the global stores.
    // role: -1 none, 0..: MFMA worker index, 16 + i: store worker i.
    if (wave == 0 && lane < NW) {
        const int sm = wsimd[lane];
        int rank = 0; bool shared = false;
        for (int w = 0; w < NW; ++w) { if (w < lane && wsimd[w] == sm) ++rank; if (w < NEW && wsimd[w] == sm) shared = true; }
        int kind = (lane < NEW || shared) ? -1 : (rank < 2 ? 0 : 1);
        if (__ballot(kind == 0) == 0ull) kind = (lane < NEW) ? -1 : (lane == NEW ? 0 : 1);   // no SIMD without an eliminating wave
        const unsigned long long mm = __ballot(kind == 0), ms = __ballot(kind == 1);
        const unsigned long long below = (1ull << lane) - 1ull;
        wrole[lane] = kind < 0 ? -1 : (kind == 0 ? __popcll(mm & below) : 16 + __popcll(ms & below));
        if (lane == 0) { nrole[0] = __popcll(mm); nrole[1] = __popcll(ms); }
    } else if (wave >= 1 && wave < NSB && lane < 32) {
        // the R tiles of block column j as LDS byte offsets {C tile, row operand, its row stride, column operand}; entries
        // past the last tile (always including 31) are a harmless dummy
        const int j = wave - 1, c0 = j * SB, nb16 = (NB - c0 - SB) / SB;
        const int nR = nb16 * (nb16 - 1) / 2 + (j + 1) * (nb16 - 1);
        TileAddr x = {0, 0, LS, &Ls[0][0]};
        if (lane < nR) tile_decode<false>(Ls, xid[j & 1], j, lane, x.crow, x.ccol, x.ap, x.astride);
        ttab[j][lane][0] = (int)lds_off(&Ls[x.crow][x.ccol]);
        ttab[j][lane][1] = (int)lds_off(x.ap);
        ttab[j][lane][2] = x.astride * 8;
        ttab[j][lane][3] = (int)lds_off(&Ls[x.ccol][c0]);
    }
#pragma unroll
    for (int i = 0; i < NB * NB / (2 * DT); ++i) {
        const int e = tid * 2 + i * 2 * DT, r = e >> 7, c = e & 127;
        d2 v = img[i];
        if (c > r) v[0] = 0.0;
        if (c + 1 > r) v[1] = 0.0;
        *(d2*)&Ls[r][c] = v;
    }
    lds_barrier();
    const int role = __builtin_amdgcn_readfirstlane(wrole[wave]), n_mw = __builtin_amdgcn_readfirstlane(nrole[0]),
              n_sw = __builtin_amdgcn_readfirstlane(nrole[1]);
    STAMP(0);

    // 1/pivot and 1/sqrt(pivot) of block column jb for the updates and the stores (16 lanes of one wave); a non-positive
    // pivot is reported here (lowest index first).
    auto pivot_scales = [&](int jb) {
        const int c0 = jb * SB;
        const double p = dout[jb & 1][lane & 15][lane & 15];
        const unsigned long long bad = __ballot(lane < SB && !(p > 0.0));
        if (bad != 0ull && lane == 0) atomicCAS((int*)info, 0, global_row0 + c0 + (int)__ffsll((long long)bad));
        if (lane < SB) dinv[c0 + lane] = rsqrt_nr(p);
    };
    // Everything that becomes final with block column jb -- columns c0..c0+15 of L, of L^-T (rows 0..c0+15) and rows
    // c0..c0+15 of L^-1 --, scaled by 1/sqrt(pivot) of its column on the way out, as fire-and-forget global stores by
    // threads [0, nt).  U(i, c) below = row i of L^-T, unscaled.  Every section has at most 1024 items: two
    // predicated items per thread and section, written so that the six LDS reads of a thread are in flight together.
    auto store_block_column = [&](int jb, int t0, int nt) {
        // nt is a multiple of 64.  A thread keeps its column pair (or its pair of rows of the inverse) for a whole section, so
        // its scale factors are loaded once and both addresses advance by constants: ~10 instructions per 16 bytes stored (a
        // wave issues one instruction every ~4-5 cycles; with the index arithmetic redone per item the stores of a block
        // column took longer than its tile updates).  The empty asm keeps the per-lane constants from being hoisted out of
        // the block-column loop, where they would be live through the elimination and spill it.
        asm volatile("" : "+v"(t0));
        const int c0 = jb * SB;
        const double (*xb)[XS3] = xid[jb & 1];
        const double (*db)[SB] = dout[jb & 1];
        const int c = 2 * (t0 & 7), r0 = t0 >> 3, rstep = nt >> 3;
        const d2 rs = *(const d2*)&dinv[c0 + c];
        {   // L[c0 + rr][c0 + c .. c+1]: the diagonal sub-block from the staging block (zero above the diagonal) ...
            for (int rr = r0; rr < SB; rr += rstep) {
                const d2 x = *(const d2*)&db[rr][c];
                d2 w;
                w[0] = (c > rr) ? 0.0 : x[0] * rs[0];
                w[1] = (c + 1 > rr) ? 0.0 : x[1] * rs[1];
                *(d2*)(Mblk + (long long)(c0 + rr) * ld + c0 + c) = w;
            }
            // ... the rows below it from the image
            int rr = r0 < SB ? r0 + ((SB - r0 + rstep - 1) / rstep) * rstep : r0;      // first row >= SB of this thread
            const double* src = &Ls[c0 + rr][c0 + c];
            double* dst = Mblk + (long long)(c0 + rr) * ld + c0 + c;
            const long long dstep = (long long)rstep * ld;
            for (; rr < NB - c0; rr += rstep, src += rstep * LS, dst += dstep) {
                const d2 x = *(const d2*)src;
                *(d2*)dst = (d2){x[0] * rs[0], x[1] * rs[1]};
            }
        }
        {   // LinvT[i][c0 + c .. c+1] = U(i, .) . rs: rows i < c0 from the image (all of them right of the diagonal), ...
            const double* src = &Ls[r0][c0 + c];
            double* dst = LinvT + (long long)r0 * ldinv + c0 + c;
            const long long dstep = (long long)rstep * ldinv;
            int i = r0;
            for (; i < c0; i += rstep, src += rstep * LS, dst += dstep) {
                const d2 x = *(const d2*)src;
                *(d2*)dst = (d2){x[0] * rs[0], x[1] * rs[1]};
            }
            // ... rows c0..c0+15 from the identity-row block (zeros left of the diagonal included; pairs wholly left of it skipped)
            for (; i < c0 + SB; i += rstep, dst += dstep) {
                if (c0 + c + 1 >= i) {
                    const d2 x = *(const d2*)&xb[i - c0][c];
                    *(d2*)dst = (d2){x[0] * rs[0], x[1] * rs[1]};
                }
            }
        }
        {   // Linv[c0 + r][i .. i+1] = U(i, c0 + r) . rs[r], i <= c0 + r: a thread keeps its pair i (lanes along a row of Linv)
            // and walks the rows r; U(i, .) comes from the image for i < c0, from the identity-row block otherwise (c0 is even, so a
            // pair never straddles the two; zero above the diagonal comes from the block)
            const int i = 2 * (t0 & 63), rw0 = t0 >> 6, rwstep = nt >> 6;
            if (i < c0 + SB) {
                const bool img = i < c0;
                const double* src = img ? &Ls[i][c0 + rw0] : &xb[i - c0][rw0];
                const int s1 = img ? LS : XS3;
                double* dst = Linv + (long long)(c0 + rw0) * ldinv + i;
                const long long dstep = (long long)rwstep * ldinv;
                for (int r = rw0; r < SB; r += rwstep, src += rwstep, dst += dstep) {
                    if (i <= c0 + r) {
                        const double rs1 = dinv[c0 + r];
                        *(d2*)dst = (d2){src[0] * rs1, src[s1] * rs1};
                    }
                }
            }
        }
    };
    // The work of block column k that runs beside E(k+1): the stores on the store workers; the R tiles on the MFMA workers,
    // one accumulation chain at a time (a dependent v_mfma_f64_16x16x4 issues every ~85 cycles, the pipe's limit is ~75), the
    // operands of the next tile loaded before the MFMAs of the current one are issued (the LDS reads of a tile take
    // 300-500 cycles for a wave on its own).
    auto work_beside = [&](int k) {
        const int c0 = k * SB, nb16 = (NB - c0 - SB) / SB;
        const int nR = nb16 * (nb16 - 1) / 2 + (k + 1) * (nb16 - 1);
        if (role >= 16) {
            // above the MFMA workers of the same SIMD: the few v_mul_f64 of the stores need the double-precision pipe the MFMAs
            // keep busy, and at equal priority they only got it when both workers happened to be between tiles (the stores of a
            // block column took ~5000 cycles)
            __builtin_amdgcn_s_setprio(2);
            store_block_column(k, (role - 16) * 64 + lane, n_sw * 64);
            __builtin_amdgcn_s_setprio(0);
            return;
        }
        // Two-stage pipeline per worker: right behind the first MFMA of tile i come the operand loads of tile i+1 and the read
        // of the table entry (LDS offsets, built by wave 9 during P) of tile i+2; the ~340 cycles of the four dependent
        // MFMAs cover them.  A wave issues at most one instruction every ~4 cycles whatever its kind, so what a tile costs
        // beside its MFMAs is its instruction count: ~110 with the tile decode inline (820 cycles per tile measured), ~35 now.
        // The loads are inline asm: the compiler's own s_waitcnt in a loop was lgkmcnt(0) right in front of the first MFMA,
        // prefetch included.  Every wait here is lgkmcnt(0) at the top of a stage (no counting); tile_wait ties the registers
        // to it, and the two register sets alternate (loop unrolled twice) so that nothing in flight is ever copied.
        d2 pva, pvb;
        {
            const unsigned ap_ = lds_off(&dinv[c0 + 4 * fq]);
            asm volatile("ds_read_b128 %0, %1" : "=v"(pva) : "v"(ap_));
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(pvb) : "v"(ap_));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pva), "+v"(pvb));
        }
        const double pv[4] = {pva[0], pva[1], pvb[0], pvb[1]};
        typedef int i4 __attribute__((ext_vector_type(4)));
        const int pr = 4 * (fr & 3) + (fr >> 2);
        const unsigned lane_c = fr * (LS * 8) + 32 * fq, lane_b = pr * (LS * 8) + 32 * fq, lane_a = 32 * fq;
        const unsigned tab0 = lds_off(&ttab[k][0][0]);
        auto entry = [&](int t, i4& e) {                         // a tile past the end reads the dummy entry 31
            const unsigned a_ = tab0 + 16 * (t < nR ? t : 31);
            asm volatile("ds_read_b128 %0, %1" : "=v"(e) : "v"(a_));
        };
        auto loads = [&](const i4& e, TileRegs2& g, unsigned& ac) {
            ac = e[0] + lane_c;
            const unsigned aa = e[1] + fr * e[2] + lane_a, ab = e[3] + lane_b;
            asm volatile("ds_read_b128 %0, %1" : "=v"(g.c[0]) : "v"(ac));
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(g.c[1]) : "v"(ac));
            asm volatile("ds_read_b128 %0, %1" : "=v"(g.fa[0]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(g.fa[1]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1" : "=v"(g.fb[0]) : "v"(ab));
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(g.fb[1]) : "v"(ab));
        };
        // stage: tile t in (gc, acc) is complete after the wait, en is the entry of tile t + n_mw; issues the loads of that
        // tile into (gn, acn) and the entry read of tile t + 2 n_mw into ef
        auto stage = [&](int t, TileRegs2& gc, unsigned acc, i4& en, TileRegs2& gn, unsigned& acn, i4& ef) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gc.c[0]), "+v"(gc.c[1]), "+v"(gc.fa[0]), "+v"(gc.fa[1]), "+v"(gc.fb[0]), "+v"(gc.fb[1]), "+v"(en));
            d4 c = (d4){gc.c[0][0], gc.c[0][1], gc.c[1][0], gc.c[1][1]};
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(gc.fb[0][0] * pv[0], -(gc.fa[0][0] * pv[0]), c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            loads(en, gn, acn);
            entry(t + 2 * n_mw, ef);
            __builtin_amdgcn_sched_barrier(0);
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(gc.fb[0][1] * pv[1], -(gc.fa[0][1] * pv[1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(gc.fb[1][0] * pv[2], -(gc.fa[1][0] * pv[2]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(gc.fb[1][1] * pv[3], -(gc.fa[1][1] * pv[3]), c, 0, 0, 0);
            const d2 lo = (d2){c[0], c[1]}, hi = (d2){c[2], c[3]};
            // 18 wait states between the last MFMA and a read of its result: the hazard recognizer does not look into inline asm
            asm volatile("s_nop 15\n\ts_nop 1\n\tds_write_b128 %0, %1" :: "v"(acc), "v"(lo) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:16" :: "v"(acc), "v"(hi) : "memory");
        };
        int t = role;
        if (t < nR) {
            i4 ea, eb;
            TileRegs2 ga, gb;
            unsigned aca, acb;
            entry(t, ea);
            entry(t + n_mw, eb);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ea), "+v"(eb));
            loads(ea, ga, aca);
            for (;;) {
                stage(t, ga, aca, eb, gb, acb, ea);  t += n_mw;  if (t >= nR) break;
                stage(t, gb, acb, ea, ga, aca, eb);  t += n_mw;  if (t >= nR) break;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ea), "+v"(eb));
            tile_wait<0>(ga); tile_wait<0>(gb);
        }
        if (n_sw == 0) store_block_column(k, role * 64 + lane, n_mw * 64);
    };

    // One workgroup barrier per block column.  Between two barriers, with X(k-1) known:
    //   every wave < 8 first updates its tile of block column k (P(k-1)) and counts it in pdone[k-1];
    //   the eliminating waves then wait for all 8 tiles (they read every row of block column k) and run E(k);
    //   wave 8 computes the scales of block k-1 and posts ready[k-1];
    //   MFMA workers and store workers wait for that and do R(k-1) / the stores of block column k-1.
    // R(k-1) touches only columns right of block k and reads block column k-1, P(k-1) writes block column k: they run
    // side by side, and so do the stores (block column k-1, final).  An LDS operation of a wave completes before its next one,
    // so a tile is in place when its counter increment lands.
    auto post = [&](int* f) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) atomicAdd(f, 1);
    };
    auto await = [&](int* f, int target) {
        while (__atomic_load_n(f, __ATOMIC_RELAXED) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto p_tile = [&](int j) {                               // tile `wave` of P(j): block column j+1 updated by block column j
        double pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = dout[j & 1][4 * fq + u][4 * fq + u];
        TileAddr a0;
        tile_decode<true>(Ls, xid[j & 1], j, wave, a0.crow, a0.ccol, a0.ap, a0.astride);
        update_tile_rawpiv(Ls, a0, j * SB, pv, fr, fq);
        post(&pdone[j]);
    };
    for (int k = 0; k < NSB; ++k) {
        if (k > 0 && wave < 8) p_tile(k - 1);
        if (k == 1) WSTAMP(32);
        if (wave < NEW) {
            if (k > 0) await(&pdone[k - 1], 8);
            if (k == 1) STAMP(2);
            __builtin_amdgcn_s_setprio(3);
            eliminate16<NARR>(Ls, xid[k & 1], dout[k & 1], eye, k, wave, lane);
            __builtin_amdgcn_s_setprio(0);
        } else if (k > 0) {
            if (wave == 8) {      // ahead of the P tiles that share its SIMD: every worker waits for these sixteen numbers
                __builtin_amdgcn_s_setprio(3);
                pivot_scales(k - 1);
                post(&ready[k - 1]);
                __builtin_amdgcn_s_setprio(0);
            }
            if (role >= 0) {
                await(&ready[k - 1], 1);
                work_beside(k - 1);
            }
        }
        if (k == 1) WSTAMP(16);
        lds_barrier();
        if (k == 0) STAMP(1);
        if (k == 1) STAMP(3);
    }
    STAMP(6);
    if (wave == 8) pivot_scales(NSB - 1);
    lds_barrier();
    store_block_column(NSB - 1, tid, DT);
    STAMP(7);
#undef STAMP
#undef WSTAMP
}

long long* g_diag_stamps = nullptr;  // debug: device buffer of 8 cycle stamps for block 0 (scripts/)
// Two-level right-looking factorisation.  Inside an outer panel of OUTER 128-blocks the steps are
//   diag(j) -> panel solve(j) -> update of the REST OF THE OUTER PANEL only (K = 128, few tiles)
// and the big trailing update runs once per outer panel with K = OUTER*128: 4x fewer read-modify-write
// passes over the trailing matrix (at K = 128 that update is bound by the C-tile traffic, 16 flop/B).
constexpr int OUTER = POTRF_OUTER;

hipError_t potrf_clear_info(int32_t* info, hipStream_t st, const Batch& bt) {
    // (a finished LP of a batch has its info word cleared too: its status record already holds the value)
    info = (int32_t*)((char*)info + (size_t)bt.first * (size_t)bt.stride);
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
        hipLaunchKernelGGL(potrf_diag_kernel<2>, dim3(1, 1, bt.count), dim3(DT), 0, st, diag, (long long)ld, linv,
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
            // rows j+1..nb x columns j+1..J1-1 -= L[rows, j] . L[cols, j]^T  (rectangular grid of tiles; the few tiles
            // above the diagonal are computed too and never read)
            GemmArgs c{};
            c.P = panel; c.ldp = ld; c.Q = panel; c.ldq = ld; c.s = nullptr;
            c.C = M + (o + NB) * ld + (o + NB); c.ldc = ld; c.K = NB; c.alpha = -1.0; c.beta = 1.0;
            // 32x32 tiles: 16x as many workgroups as 128x128 ones, 32 MFMAs per wave; the launch is one round of tiles on the
            // chain (factorisation at m = 4096: 1998 us with 64x64 tiles, 1942 with 32x64, 1903 with 32x32)
            c.tile_edge = 3232; c.tiles_lower = 0; c.ntj = 4 * ncols; c.ntiles = (4 * rem) * (4 * ncols);
            c.tile_list = nullptr; c.diag_pad_from = -1; c.ws = nullptr; c.nwg = c.ntiles; c.batch = bt;
            e = launch_gemm_nt(c, st);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

// Right-looking trailing update behind the outer panel [J0, J1): A22 -= L21 . L21^T with K = (J1 - J0) * 128, restricted to
// the block columns [C0, C1) of the trailing matrix (C0 >= J1; rows >= the column's own block).
//   whole update: C0 = J1, C1 = nb.  Look-ahead: [J1, J1 + OUTER) first -- all the next outer panel reads -- and the rest on a
//   side stream beside that panel's chain.
static hipError_t trailing_update_columns(double* M, int64_t ld, int nb, hipStream_t st, const Batch& bt, int J0, int J1,
                                          int C0, int C1) {
    if (C0 >= C1) return hipSuccess;
    const int64_t oJ0 = (int64_t)J0 * NB, oC0 = (int64_t)C0 * NB;
    const int remT = nb - C0;                    // block rows from C0 down
    GemmArgs u{};
    u.s = nullptr; u.ldp = ld; u.ldq = ld; u.ldc = ld; u.K = (J1 - J0) * NB; u.alpha = -1.0; u.beta = 1.0;
    u.tile_list = nullptr; u.diag_pad_from = -1; u.ws = nullptr; u.batch = bt;
    u.P = M + oC0 * ld + oJ0; u.Q = u.P; u.C = M + oC0 * ld + oC0;
    if (C1 < nb) {
        // a band of columns: rectangular grid of 32x32 tiles, rows C0..nb x columns C0..C1 (the few tiles above the diagonal
        // are computed too and never read); it sits on the chain, and a 32x32 tile is the shortest MFMA chain there is
        u.tile_edge = 3232; u.tiles_lower = 0; u.ntj = 4 * (C1 - C0); u.ntiles = (4 * remT) * (4 * (C1 - C0));
    } else {
        u.tiles_lower = 1; u.ntj = 0;
        // up to 16 trailing blocks (at most one round of 64x64 tiles) 32x32 tiles: the launch lasts as long as the busiest CU's
        // tiles, and a quarter-size tile is a quarter-length MFMA chain (m = 4096: 1889 -> 1858 us, m = 2048: 772 -> 745);
        // 64x64 tiles (4 workgroups per CU) until the 128x128 ones would fill the chip's 512 slots about twice
        if (remT <= 16)                        { u.tile_edge = 3232; u.ntiles = (4 * remT) * (4 * remT + 1) / 2; }
        else if (remT * (remT + 1) / 2 < 1024) { u.tile_edge = 64; u.ntiles = (2 * remT) * (2 * remT + 1) / 2; }
        else                                   { u.tile_edge = 128; u.ntiles = remT * (remT + 1) / 2; }
    }
    u.nwg = u.ntiles;
    return launch_gemm_nt(u, st);
}
hipError_t potrf_trailing_update(double* M, int64_t ld, int mp, hipStream_t st, const Batch& bt, int J0, int J1) {
    return trailing_update_columns(M, ld, mp / NB, st, bt, J0, J1, J1, mp / NB);
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
                        const Batch& bt, const PotrfLookahead* la, bool clear_info) {
    hipError_t e = clear_info ? potrf_clear_info(info, st, bt) : hipSuccess;   // (the solver's iteration keeps the word clean itself)
    if (e != hipSuccess) return e;
    const int nb = mp / NB;
    // Look-ahead (one LP, enough trailing matrix for it to matter): behind outer panel p the block columns of panel p+1 are
    // updated on the chain stream and the chain goes on; the rest of the update runs on the side stream beside it and has to
    // be complete only before panel p+1's own trailing update touches the same columns.  On from m = 4096 (round 3; round 2
    // had it opt-in): measured at m = 4096 1910 -> 1855 us (round 2), 1866 -> 1820 us and 209.5 -> 214.2 it/s at C3 (round 3,
    // same box, alternating runs) with 8 CUs per XCC kept free for the chain, 1870 with 4, 1890 with 2, 1905 with a
    // low-priority unmasked side stream; at m = 2048 750 -> 775 whatever the setting, hence the threshold (LPIPM_LOOKAHEAD=1
    // lowers it to 12 blocks, =0 switches the side stream off).  Trace: the chain's part of the update is
    // one round of K = 512 tiles (45 us where the whole update takes 105), and the chain's next panel solve and inner update
    // take 2-4x as long beside the side stream's workgroups (26 and 46 us instead of 10): 270 us per outer panel against 275.
    const int npanel = (nb + OUTER - 1) / OUTER;
    const bool ahead = la && la->side && bt.count == 1 && (int)la->ev_chain.size() >= npanel && nb >= 3 * OUTER && nb >= la->min_nb;
    int pending = -1;                            // panel whose rest-update the chain stream has not waited for yet
    for (int J0 = 0, pnl = 0; J0 < nb; J0 += OUTER, ++pnl) {
        const int J1 = J0 + OUTER < nb ? J0 + OUTER : nb;
        if ((e = potrf_panel_chain(M, ld, mp, plan, info, st, bt, J0, J1)) != hipSuccess) return e;
        if (J1 >= nb) break;
        const int J2 = J1 + OUTER < nb ? J1 + OUTER : nb;
        if (ahead && J2 < nb) {
            if (pending >= 0 && (e = hipStreamWaitEvent(st, la->ev_rest[pending], 0)) != hipSuccess) return e;
            if ((e = trailing_update_columns(M, ld, nb, st, bt, J0, J1, J1, J2)) != hipSuccess) return e;
            // the rest starts once the chain stream's part is done: side by side the two halve each other and the chain waits
            // as long as for the whole update (trace: 103 us instead of 12)
            if ((e = hipEventRecord(la->ev_chain[pnl], st)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(la->side, la->ev_chain[pnl], 0)) != hipSuccess) return e;
            if ((e = trailing_update_columns(M, ld, nb, la->side, bt, J0, J1, J2, nb)) != hipSuccess) return e;
            if ((e = hipEventRecord(la->ev_rest[pnl], la->side)) != hipSuccess) return e;
            pending = pnl;
        } else {
            if (pending >= 0) { if ((e = hipStreamWaitEvent(st, la->ev_rest[pending], 0)) != hipSuccess) return e; pending = -1; }
            if ((e = trailing_update_columns(M, ld, nb, st, bt, J0, J1, J1, nb)) != hipSuccess) return e;
        }
    }
    if (pending >= 0 && (e = hipStreamWaitEvent(st, la->ev_rest[pending], 0)) != hipSuccess) return e;
    return potrf_superblock_inverses(plan, st, bt);
}

}  // namespace lpipm
