// kernels_gemm.hip -- fp64 MFMA "NT" GEMM for gfx950 (CDNA4), the dominant kernel of the hot path.
//
//   C(ti,tj) = beta*C + alpha * sum_k P[ti*128 + r][k] * s[k] * Q[tj*128 + c][k]
//
// Replaces newton_equations.rs:54-57 (`A.dot(&(Dinv[:,None] * A.t()))`): the n x m scaled
// temporary is never materialised -- s = x/z is applied to the Q panel while it is staged into LDS
// -- and only the lower-triangular tiles are computed (the reference forms the full square).
// The same kernel is the Cholesky trailing update (alpha=-1, beta=1) and TRSM-as-GEMM.
//
// Design (MI355X: 256 CUs, wave64, 160 KB LDS/CU, v_mfma_f64_16x16x4_f64):
//   * 128x128 output tile per 256-thread workgroup = 2x2 waves of 64x64 = 4x4 MFMA tiles each:
//     64 fp64 accumulators per lane (128 VGPRs), 2 workgroups per CU (2 waves per SIMD) so one
//     wave's LDS/barrier stalls hide under the other's MFMAs.
//   * both operands are row-major with K contiguous, which is exactly the A/B fragment shape of
//     the 16x16x4 MFMA (lane l holds X[l&15][k = l>>4]); K is permuted so that each lane reads
//     two consecutive k (one ds_read_b128) per pair of MFMAs.
//   * k-tiles of 16 are register-staged (global_load_dwordx4: 8 lanes x 16 B = one full 128-B line
//     per row), written to a double-buffered padded LDS image (row stride 18 doubles: the 16
//     rows x 2 k-groups of a 32-lane LDS phase land on distinct banks), one barrier per k-tile.
//   * stream-K: ntiles*KT k-tile iterations are split evenly over the launched workgroups
//     (528 lower tiles at m=4096 do not divide over 512 resident workgroups); a workgroup's
//     partial first/last tile goes to a slab and a second pass adds the slabs of a tile in
//     workgroup order -- deterministic, no atomics.
//   * workgroups are renumbered so that the 64 that share an XCD (and its L2) work on one
//     8x8 super-block of tiles: 16 row panels of A feed 64 tiles.
#include "lpipm_internal.hpp"
#include <cstdlib>

namespace lpipm {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int LDS_STRIDE = BK + 2;  // doubles per LDS row (144 B, keeps 16-B alignment)
constexpr int SK_CHUNK = 16;        // k-tiles per dynamically claimed stream-K chunk

struct GemmK {
    const double* P; long long ldp;
    const double* Q; long long ldq;
    const double* s;
    double* C; long long ldc;
    int KT;
    double alpha, beta;
    int ntiles, tiles_lower, ntj;
    const int2* tile_list;
    int diag_pad_from;
    double* ws;
    int nwg;
    unsigned int* sk_claim;   // see GemmArgs
    BatchK bk;
};
// LP blockIdx.z of a lockstep batch: per-LP pointers shifted (the tile list is shared)
__device__ __forceinline__ GemmK batch_shift(const GemmK& p0) {
    GemmK p = p0;
    p.P = batch_ptr(p0.P, p0.bk); p.Q = batch_ptr(p0.Q, p0.bk); p.s = batch_ptr(p0.s, p0.bk);
    p.C = batch_ptr(p0.C, p0.bk); p.ws = batch_ptr(p0.ws, p0.bk);
    return p;
}

__device__ __forceinline__ int xcd_remap(int b, int n) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range.
    const int xcd = b & 7, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ void tile_coords(const GemmK& p, int t, int& ti, int& tj) {
    if (p.tile_list) {
        const int2 c = p.tile_list[t];
        ti = c.x; tj = c.y;
    } else if (p.tiles_lower) {
        int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        ti = i; tj = t - i * (i + 1) / 2;
    } else {
        ti = t / p.ntj; tj = t - ti * p.ntj;
    }
}

__device__ __forceinline__ long long wg_begin(long long g, long long total, int nwg) {
    return (g * total) / nwg;
}
// workgroup that owns k-tile iteration `it`
__device__ __forceinline__ int wg_owner(long long it, long long total, int nwg) {
    return (int)(((it + 1) * (long long)nwg - 1) / total);
}


// acc(128x128 tile, 64 doubles per lane) += sum over k-tiles [kb, ke) of P-panel . (s o Q-panel)^T
// Pp / Qp point at this thread's first staging element (row srow, column scol of the panels).
// MTM x MTN = 16x16 MFMA tiles per wave (2x2 waves per workgroup): workgroup tile = 32*MTM x 32*MTN.
//   4x4 -> 128x128 (throughput launches), 2x2 -> 64x64 and 1x4 -> 32x128 (latency-bound launches)
template <bool SCALE, int MTM, int MTN>
__device__ __forceinline__ void tile_mainloop(double (*ldsA)[32 * MTM][LDS_STRIDE], double (*ldsB)[32 * MTN][LDS_STRIDE],
                                              const double* __restrict__ Pp, long long ldp,
                                              const double* __restrict__ Qp, long long ldq,
                                              const double* __restrict__ s, int kb, int ke, d4 (&acc)[MTM][MTN],
                                              int srow, int scol, int wr, int wc, int fr, int fq) {
    d2 sa[MTM], sb[MTN], sv = (d2){1.0, 1.0};
    auto gload = [&](int kt) {
        const long long ko = (long long)kt * BK;
#pragma unroll
        for (int r = 0; r < MTM; ++r) sa[r] = *(const d2*)(Pp + (long long)(32 * r) * ldp + ko);
#pragma unroll
        for (int r = 0; r < MTN; ++r) sb[r] = *(const d2*)(Qp + (long long)(32 * r) * ldq + ko);
        if (SCALE) sv = *(const d2*)(s + ko + scol);
    };
    // the scale is applied here, after the MFMAs of the current k-tile: multiplying right after the
    // loads would make the wave wait for the prefetch it has just issued
    auto lstore = [&](int buf) {
#pragma unroll
        for (int r = 0; r < MTM; ++r) *(d2*)&ldsA[buf][srow + 32 * r][scol] = sa[r];
#pragma unroll
        for (int r = 0; r < MTN; ++r) *(d2*)&ldsB[buf][srow + 32 * r][scol] = SCALE ? sb[r] * sv : sb[r];
    };
    gload(kb);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kb; kt < ke; ++kt) {
        const bool more = kt + 1 < ke;
        if (more) gload(kt + 1);
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            d2 a[MTM], b[MTN];
#pragma unroll
            for (int mi = 0; mi < MTM; ++mi)
                a[mi] = *(const d2*)&ldsA[cur][wr * (16 * MTM) + mi * 16 + fr][round * 8 + fq * 2];
#pragma unroll
            for (int nj = 0; nj < MTN; ++nj)
                b[nj] = *(const d2*)&ldsB[cur][wc * (16 * MTN) + nj * 16 + fr][round * 8 + fq * 2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
                    for (int nj = 0; nj < MTN; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
}

// C tile <- beta*C + alpha*acc.  C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15,
// row = (lane>>4) + 4*reg.  Per-lane base pointer + wave-uniform row offsets keep the address math
// in SGPRs.  cb = &C[tile_row0 + wr*16*MTM + fq][tile_col0 + wc*16*MTN + fr].
template <int MTM, int MTN>
__device__ __forceinline__ void tile_store(double* cb, long long ldc, const d4 (&acc)[MTM][MTN], double alpha,
                                           double beta, bool pad_diag, int row0, int diag_pad_from, int fr, int fq,
                                           int diag_delta = 0) {   // (wave's column offset - row offset) inside the tile
    if (beta != 0.0) {   // wave-uniform.  All C values of a 16-row block are requested before any is used.
#pragma unroll
        for (int mi = 0; mi < MTM; ++mi) {
            double cv[4][MTN];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nj = 0; nj < MTN; ++nj) cv[r][nj] = cb[(long long)(mi * 16 + 4 * r) * ldc + nj * 16];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nj = 0; nj < MTN; ++nj)
                    cb[(long long)(mi * 16 + 4 * r) * ldc + nj * 16] = fma(alpha, acc[mi][nj][r], beta * cv[r][nj]);
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* rp = cb + (long long)(mi * 16 + 4 * r) * ldc;
#pragma unroll
            for (int nj = 0; nj < MTN; ++nj) {
                double v = alpha * acc[mi][nj][r];
                if (pad_diag && (mi - nj) * 16 + fq + 4 * r - fr == diag_delta && row0 + mi * 16 + 4 * r >= diag_pad_from) v = 1.0;
                rp[nj * 16] = v;
            }
        }
}

#define TILE_THREAD_IDS                                                    \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;         \
    const int wr = wave >> 1, wc = wave & 1;                               \
    const int fr = lane & 15, fq = lane >> 4;                              \
    const int srow = tid >> 3;  /* staging: 32 rows per pass, 8 lanes per 128-B row segment */ \
    const int scol = (tid & 7) * 2;

// A.D.A^T (also usable unscaled): data-parallel + stream-K hybrid.
//   phase 1: the first ntiles_dp = floor(ntiles/nwg)*nwg tiles, one whole tile per workgroup per
//            round.  Every workgroup walks k from 0 in lockstep, so the ~64 workgroups of an XCD
//            (one 8x8 super-block of tiles) hit each other's A panels in that XCD's L2.
//   phase 2: the remaining tiles' k-tile iterations are split evenly over all workgroups
//            (stream-K): the tail that would otherwise leave most CUs idle.  A workgroup's partial
//            first/last tile goes to a slab; gemm_nt_fixup_kernel adds the slabs in workgroup order.
template <bool SCALE>
__global__ __launch_bounds__(256, 2) void gemm_nt_streamk_kernel(const GemmK p0) {
    if (batch_done(p0.bk)) return;
    const GemmK p = batch_shift(p0);
    __shared__ __attribute__((aligned(16))) double ldsA[2][TILE][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][TILE][LDS_STRIDE];
    TILE_THREAD_IDS
    // batch in XCD-major layout: the workgroup's index inside its LP is blockIdx.y and the LP already sits on one XCD
    const int g = p.bk.xcd_major ? (int)blockIdx.y : xcd_remap(blockIdx.x, gridDim.x);
    const int KT = p.KT;
    const int ntiles_dp = (p.ntiles / p.nwg) * p.nwg;

    for (int tile = g; tile < ntiles_dp; tile += p.nwg) {
        int ti, tj;
        tile_coords(p, tile, ti, tj);
        d4 acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
        tile_mainloop<SCALE, 4, 4>(ldsA, ldsB, p.P + (long long)(ti * TILE + srow) * p.ldp + scol, p.ldp,
                                   p.Q + (long long)(tj * TILE + srow) * p.ldq + scol, p.ldq, p.s, 0, KT, acc, srow,
                                   scol, wr, wc, fr, fq);
        double* cb = p.C + (long long)(ti * TILE + wr * 64 + fq) * p.ldc + (tj * TILE + wc * 64 + fr);
        tile_store<4, 4>(cb, p.ldc, acc, p.alpha, p.beta, p.diag_pad_from >= 0 && ti == tj && wr == wc,
                         ti * TILE + wr * 64 + fq, p.diag_pad_from, fr, fq);
    }

    const long long total = (long long)(p.ntiles - ntiles_dp) * KT;
    long long it = wg_begin(g, total, p.nwg);
    const long long end = wg_begin(g + 1, total, p.nwg);
    bool first = true;
    while (it < end) {
        const int rt = (int)(it / KT);                       // remainder-tile index
        const int kb = (int)(it - (long long)rt * KT);
        const int ke = (int)((long long)(KT - kb) < (end - it) ? KT : kb + (end - it));
        int ti, tj;
        tile_coords(p, ntiles_dp + rt, ti, tj);
        d4 acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
        tile_mainloop<SCALE, 4, 4>(ldsA, ldsB, p.P + (long long)(ti * TILE + srow) * p.ldp + scol, p.ldp,
                                   p.Q + (long long)(tj * TILE + srow) * p.ldq + scol, p.ldq, p.s, kb, ke, acc, srow,
                                   scol, wr, wc, fr, fq);
        if (kb == 0 && ke == KT) {
            double* cb = p.C + (long long)(ti * TILE + wr * 64 + fq) * p.ldc + (tj * TILE + wc * 64 + fr);
            tile_store<4, 4>(cb, p.ldc, acc, p.alpha, p.beta, p.diag_pad_from >= 0 && ti == tj && wr == wc,
                             ti * TILE + wr * 64 + fq, p.diag_pad_from, fr, fq);
        } else {
            double* sb0 = p.ws + ((long long)(2 * g + (first ? 0 : 1))) * (TILE * TILE) +
                          (wr * 64 + fq) * TILE + wc * 64 + fr;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj) sb0[(mi * 16 + 4 * r) * TILE + nj * 16] = acc[mi][nj][r];
        }
        it += ke - kb;
        first = false;
    }
}

// ---------------------------------------------------------------------------------------------------------
// 8-wave form of the same 128x128 workgroup tile: 512 threads = 2x4 waves of 64x32 (4x2 MFMA tiles, 32 fp64
// accumulators per lane), <= 128 VGPRs, so TWO such workgroups = 4 waves per SIMD are resident per CU.
// Why: with 2 waves per SIMD a wave's non-MFMA phase of a k-tile (issue the prefetch, ds_read the fragments,
// scale + ds_write the next k-tile, barrier: ~4000 cycles under contention, s_memtime stamps) is as long as its
// partner's MFMA phase (64 x 64 cycles), so the two can only just cover each other (measured 87 % MFMA-busy);
// with 4 waves per SIMD each wave issues 32 MFMAs per k-tile and has three partners' 6144 cycles of cover.
// Same operands, LDS image, stream-K split, slabs and fix-up as gemm_nt_streamk_kernel.
template <bool SCALE>
__device__ __forceinline__ void tile_mainloop_w8(double (*ldsA)[TILE][LDS_STRIDE], double (*ldsB)[TILE][LDS_STRIDE],
                                                 const double* __restrict__ Pp, long long ldp,
                                                 const double* __restrict__ Qp, long long ldq,
                                                 const double* __restrict__ s, int kb, int ke, d4 (&acc)[4][2],
                                                 int srow, int scol, int wr, int wc, int fr, int fq) {
    d2 sa[2], sb[2], sv = (d2){1.0, 1.0};
    auto gload = [&](int kt) {
        const long long ko = (long long)kt * BK;
#pragma unroll
        for (int r = 0; r < 2; ++r) sa[r] = *(const d2*)(Pp + (long long)(64 * r) * ldp + ko);
#pragma unroll
        for (int r = 0; r < 2; ++r) sb[r] = *(const d2*)(Qp + (long long)(64 * r) * ldq + ko);
        if (SCALE) sv = *(const d2*)(s + ko + scol);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 2; ++r) *(d2*)&ldsA[buf][srow + 64 * r][scol] = sa[r];
#pragma unroll
        for (int r = 0; r < 2; ++r) *(d2*)&ldsB[buf][srow + 64 * r][scol] = SCALE ? sb[r] * sv : sb[r];
    };
    gload(kb);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kb; kt < ke; ++kt) {
        const bool more = kt + 1 < ke;
        if (more) gload(kt + 1);
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            d2 a[4], b[2];
            if (round == 0) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = *(const d2*)&ldsA[cur][wr * 64 + mi * 16 + fr][round * 8 + fq * 2];
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) b[nj] = *(const d2*)&ldsB[cur][wc * 32 + nj * 16 + fr][round * 8 + fq * 2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(2);   // the short non-MFMA phase goes first: it is what the partners wait for
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    __builtin_amdgcn_s_setprio(0);
}

template <bool SCALE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_streamk_w8_kernel(const GemmK p0) {
    if (batch_done(p0.bk)) return;
    const GemmK p = batch_shift(p0);
    __shared__ __attribute__((aligned(16))) double ldsA[2][TILE][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][TILE][LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;          // 2 x 4 waves: 64 rows x 32 columns each
    const int fr = lane & 15, fq = lane >> 4;
    const int srow = tid >> 3, scol = (tid & 7) * 2;  // staging: 64 rows per pass
    const int g = p.bk.xcd_major ? (int)blockIdx.y : xcd_remap(blockIdx.x, gridDim.x);
    const int KT = p.KT;
    const int ntiles_dp = (p.ntiles / p.nwg) * p.nwg;
    auto zero = [](d4 (&acc)[4][2]) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
    };
    auto store_tile = [&](const d4 (&acc)[4][2], int ti, int tj) {
        double* cb = p.C + (long long)(ti * TILE + wr * 64 + fq) * p.ldc + (tj * TILE + wc * 32 + fr);
        tile_store<4, 2>(cb, p.ldc, acc, p.alpha, p.beta, p.diag_pad_from >= 0 && ti == tj, ti * TILE + wr * 64 + fq,
                         p.diag_pad_from, fr, fq, wc * 32 - wr * 64);
    };
    for (int tile = g; tile < ntiles_dp; tile += p.nwg) {
        int ti, tj;
        tile_coords(p, tile, ti, tj);
        d4 acc[4][2];
        zero(acc);
        tile_mainloop_w8<SCALE>(ldsA, ldsB, p.P + (long long)(ti * TILE + srow) * p.ldp + scol, p.ldp,
                                p.Q + (long long)(tj * TILE + srow) * p.ldq + scol, p.ldq, p.s, 0, KT, acc, srow, scol, wr,
                                wc, fr, fq);
        store_tile(acc, ti, tj);
    }
    if (p.sk_claim) {
        // Dynamic stream-K: the remainder tiles' k-ranges in chunks of SK_CHUNK k-tiles, claimed through one device word.
        // The static split below hands every workgroup exactly one such chunk (at C3: 16 tiles x 512 k-tiles over 512
        // workgroups), but the data-parallel tiles do not all finish together (stamps: 1876 .. 2025 us), so the static
        // version ends 67 us after the slowest of them; claimed chunks go to whoever is free.  Slab c holds chunk c and the
        // fix-up adds a tile's slabs in chunk order: same partial sums, same order, same bits as the static split.
        __shared__ int s_claim;
        const int cpt = KT / SK_CHUNK, nchunks = (p.ntiles - ntiles_dp) * cpt;
        for (;;) {
            if (tid == 0) s_claim = (int)atomicAdd(p.sk_claim, 1u);
            __syncthreads();
            const int ch = s_claim;
            __syncthreads();
            if (ch >= nchunks) break;
            const int rt = ch / cpt, kb = (ch - rt * cpt) * SK_CHUNK;
            int ti, tj;
            tile_coords(p, ntiles_dp + rt, ti, tj);
            d4 acc[4][2];
            zero(acc);
            tile_mainloop_w8<SCALE>(ldsA, ldsB, p.P + (long long)(ti * TILE + srow) * p.ldp + scol, p.ldp,
                                    p.Q + (long long)(tj * TILE + srow) * p.ldq + scol, p.ldq, p.s, kb, kb + SK_CHUNK, acc, srow,
                                    scol, wr, wc, fr, fq);
            double* sb0 = p.ws + (long long)ch * (TILE * TILE) + (wr * 64 + fq) * TILE + wc * 32 + fr;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) sb0[(mi * 16 + 4 * r) * TILE + nj * 16] = acc[mi][nj][r];
        }
        return;
    }
    const long long total = (long long)(p.ntiles - ntiles_dp) * KT;
    long long it = wg_begin(g, total, p.nwg);
    const long long end = wg_begin(g + 1, total, p.nwg);
    bool first = true;
    while (it < end) {
        const int rt = (int)(it / KT);
        const int kb = (int)(it - (long long)rt * KT);
        const int ke = (int)((long long)(KT - kb) < (end - it) ? KT : kb + (end - it));
        int ti, tj;
        tile_coords(p, ntiles_dp + rt, ti, tj);
        d4 acc[4][2];
        zero(acc);
        tile_mainloop_w8<SCALE>(ldsA, ldsB, p.P + (long long)(ti * TILE + srow) * p.ldp + scol, p.ldp,
                                p.Q + (long long)(tj * TILE + srow) * p.ldq + scol, p.ldq, p.s, kb, ke, acc, srow, scol, wr,
                                wc, fr, fq);
        if (kb == 0 && ke == KT) {
            store_tile(acc, ti, tj);
        } else {
            double* sb0 = p.ws + ((long long)(2 * g + (first ? 0 : 1))) * (TILE * TILE) + (wr * 64 + fq) * TILE + wc * 32 + fr;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) sb0[(mi * 16 + 4 * r) * TILE + nj * 16] = acc[mi][nj][r];
        }
        it += ke - kb;
        first = false;
    }
}

// One full tile per workgroup (Cholesky trailing update, TRSM-as-GEMM): no k-split, no slabs.
// Tile coordinates are in units of (32*MTM rows, 32*MTN columns).
template <int MTM, int MTN>
__global__ __launch_bounds__(256, 2) void gemm_nt_tile_kernel(const GemmK p0) {
    if (batch_done(p0.bk)) return;
    const GemmK p = batch_shift(p0);
    constexpr int TM = 32 * MTM, TN = 32 * MTN;
    __shared__ __attribute__((aligned(16))) double ldsA[2][TM][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][TN][LDS_STRIDE];
    TILE_THREAD_IDS
    int ti, tj;
    tile_coords(p, blockIdx.x, ti, tj);
    d4 acc[MTM][MTN];
#pragma unroll
    for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
        for (int nj = 0; nj < MTN; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
    tile_mainloop<false, MTM, MTN>(ldsA, ldsB, p.P + (long long)(ti * TM + srow) * p.ldp + scol, p.ldp,
                                   p.Q + (long long)(tj * TN + srow) * p.ldq + scol, p.ldq, nullptr, 0, p.KT, acc,
                                   srow, scol, wr, wc, fr, fq);
    double* cb = p.C + (long long)(ti * TM + wr * (16 * MTM) + fq) * p.ldc + (tj * TN + wc * (16 * MTN) + fr);
    tile_store<MTM, MTN>(cb, p.ldc, acc, p.alpha, p.beta, false, 0, -1, fr, fq);
}

// K = 128 specialisation of the whole-tile kernel for the latency-bound steps of the factorisation
// (panel solve, update inside an outer panel): with 16 MFMAs per wave and k-tile the one-ahead
// prefetch of tile_mainloop cannot hide a global-load round trip, so all 8 k-tiles are requested up
// front (they fit in registers for these small tiles) and the round trip is paid once per tile.
template <int MTM, int MTN>
__global__ __launch_bounds__(256, 2) void gemm_nt_tile_k128_kernel(const GemmK p0) {
    if (batch_done(p0.bk)) return;
    const GemmK p = batch_shift(p0);
    constexpr int TM = 32 * MTM, TN = 32 * MTN, KT8 = 8;
    __shared__ __attribute__((aligned(16))) double ldsA[2][TM][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][TN][LDS_STRIDE];
    TILE_THREAD_IDS
    int ti, tj;
    tile_coords(p, blockIdx.x, ti, tj);
    const double* Pp = p.P + (long long)(ti * TM + srow) * p.ldp + scol;
    const double* Qp = p.Q + (long long)(tj * TN + srow) * p.ldq + scol;
    d2 pa[KT8][MTM], pb[KT8][MTN];
#pragma unroll
    for (int kt = 0; kt < KT8; ++kt) {
#pragma unroll
        for (int r = 0; r < MTM; ++r) pa[kt][r] = *(const d2*)(Pp + (long long)(32 * r) * p.ldp + kt * BK);
#pragma unroll
        for (int r = 0; r < MTN; ++r) pb[kt][r] = *(const d2*)(Qp + (long long)(32 * r) * p.ldq + kt * BK);
    }
    d4 acc[MTM][MTN];
#pragma unroll
    for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
        for (int nj = 0; nj < MTN; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < KT8; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int r = 0; r < MTM; ++r) *(d2*)&ldsA[buf][srow + 32 * r][scol] = pa[kt][r];
#pragma unroll
        for (int r = 0; r < MTN; ++r) *(d2*)&ldsB[buf][srow + 32 * r][scol] = pb[kt][r];
        __syncthreads();   // buffer `buf` was last read two k-tiles ago: that read finished before the previous barrier
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            d2 a[MTM], b[MTN];
#pragma unroll
            for (int mi = 0; mi < MTM; ++mi)
                a[mi] = *(const d2*)&ldsA[buf][wr * (16 * MTM) + mi * 16 + fr][round * 8 + fq * 2];
#pragma unroll
            for (int nj = 0; nj < MTN; ++nj)
                b[nj] = *(const d2*)&ldsB[buf][wc * (16 * MTN) + nj * 16 + fr][round * 8 + fq * 2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
                    for (int nj = 0; nj < MTN; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);
        }
    }
    double* cb = p.C + (long long)(ti * TM + wr * (16 * MTM) + fq) * p.ldc + (tj * TN + wc * (16 * MTN) + fr);
    tile_store<MTM, MTN>(cb, p.ldc, acc, p.alpha, p.beta, false, 0, -1, fr, fq);
}

// Grouped GEMM: every workgroup takes its own descriptor (operands, k-range, alpha): the doubling
// levels of the super-block triangular inverse are a few such launches over many small products.
__global__ __launch_bounds__(256, 2) void gemm_nt_grouped_kernel(const GemmTileDesc* __restrict__ descs, BatchK bk) {
    if (batch_done(bk)) return;
    __shared__ __attribute__((aligned(16))) double ldsA[2][TILE][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][TILE][LDS_STRIDE];
    TILE_THREAD_IDS
    GemmTileDesc d = descs[blockIdx.x];
    d.P = batch_ptr(d.P, bk); d.Q = batch_ptr(d.Q, bk); d.C = batch_ptr(d.C, bk);
    d4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
    tile_mainloop<false, 4, 4>(ldsA, ldsB, d.P + (long long)srow * d.ldp + scol, d.ldp, d.Q + (long long)srow * d.ldq + scol,
                         d.ldq, nullptr, d.kt_begin, d.kt_end, acc, srow, scol, wr, wc, fr, fq);
    double* cb = d.C + (long long)(wr * 64 + fq) * d.ldc + (wc * 64 + fr);
    tile_store<4, 4>(cb, d.ldc, acc, d.alpha, 0.0, false, 0, -1, fr, fq);
}

// The same with 64x64 output tiles (4 resident workgroups per CU): a merge stage of one LP is a few dozen
// 128x128 tiles on 512 slots and each tile's k-loop is pure latency, so quartering the tiles quarters the
// stage's duration; the triangular k-ranges are also tighter at 64 granularity.
__global__ __launch_bounds__(256, 4) void gemm_nt_grouped64_kernel(const GemmTileDesc* __restrict__ descs, BatchK bk) {
    if (batch_done(bk)) return;
    __shared__ __attribute__((aligned(16))) double ldsA[2][64][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][64][LDS_STRIDE];
    TILE_THREAD_IDS
    GemmTileDesc d = descs[blockIdx.x];
    d.P = batch_ptr(d.P, bk); d.Q = batch_ptr(d.Q, bk); d.C = batch_ptr(d.C, bk);
    d4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
    tile_mainloop<false, 2, 2>(ldsA, ldsB, d.P + (long long)srow * d.ldp + scol, d.ldp, d.Q + (long long)srow * d.ldq + scol,
                               d.ldq, nullptr, d.kt_begin, d.kt_end, acc, srow, scol, wr, wc, fr, fq);
    double* cb = d.C + (long long)(wr * 32 + fq) * d.ldc + (wc * 32 + fr);
    tile_store<2, 2>(cb, d.ldc, acc, d.alpha, 0.0, false, 0, -1, fr, fq);
}

// Adds the partial slabs of every stream-K (remainder) tile whose k-range was split, in workgroup
// order.  grid = remainder tiles x FIX_SPLIT: a tile can have dozens of slabs, so its 16K elements are
// spread over FIX_SPLIT workgroups (8 rows each) to keep this pass off the critical path.
constexpr int FIX_SPLIT = 16;
__global__ __launch_bounds__(256) void gemm_nt_fixup_kernel(const GemmK p0) {
    if (batch_done(p0.bk)) return;
    const GemmK p = batch_shift(p0);
    const int KT = p.KT;
    const int ntiles_dp = (p.ntiles / p.nwg) * p.nwg;
    const int bx = p.bk.xcd_major ? (int)blockIdx.y : (int)blockIdx.x;
    const int rt = bx / FIX_SPLIT, chunk = bx % FIX_SPLIT;
    const long long total = (long long)(p.ntiles - ntiles_dp) * KT;
    const long long it0 = (long long)rt * KT, it1 = it0 + KT;
    const bool dyn = p.sk_claim != nullptr;          // slabs [rt*cpt, (rt+1)*cpt), one per claimed chunk
    const int cpt = KT / SK_CHUNK;
    const int g_lo = dyn ? rt * cpt : wg_owner(it0, total, p.nwg), g_hi = dyn ? rt * cpt + cpt - 1 : wg_owner(it1 - 1, total, p.nwg);
    if (g_lo == g_hi) return;  // one workgroup computed the whole tile and wrote it directly
    int ti, tj;
    tile_coords(p, ntiles_dp + rt, ti, tj);
    constexpr int PER = TILE * TILE / FIX_SPLIT;
    for (int e = chunk * PER + threadIdx.x * 2; e < (chunk + 1) * PER; e += 512) {
        d2 sum = (d2){0.0, 0.0};
        for (int g = g_lo; g <= g_hi; ++g) {
            const int slot = wg_begin(g, total, p.nwg) >= it0 ? 0 : 1;
            sum += *(const d2*)(p.ws + (dyn ? (long long)g : (long long)(2 * g + slot)) * (TILE * TILE) + e);
        }
        const int r = e / TILE, c = e - r * TILE;
        const int row = ti * TILE + r, col = tj * TILE + c;
        double* cp = p.C + (long long)row * p.ldc + col;
        d2 v = p.alpha * sum;
        if (p.beta != 0.0) v += p.beta * (*(const d2*)cp);
        if (p.diag_pad_from >= 0) {
            if (row == col && row >= p.diag_pad_from) v[0] = 1.0;
            if (row == col + 1 && row >= p.diag_pad_from) v[1] = 1.0;
        }
        *(d2*)cp = v;
    }
}

int gemm_streamk_nwg(int ntiles, int KT, int num_cu) {
    const long long total = (long long)ntiles * KT;
    long long nwg = 2LL * num_cu;              // 2 resident workgroups per CU
    if (nwg > total / 8) nwg = total / 8;      // at least 8 k-tiles (one 128-deep slice) each
    if (nwg < 1) nwg = 1;
    return (int)nwg;
}

hipError_t launch_gemm_nt(const GemmArgs& a, hipStream_t st) {
    GemmK k;
    k.P = a.P; k.ldp = a.ldp; k.Q = a.Q; k.ldq = a.ldq; k.s = a.s;
    k.C = a.C; k.ldc = a.ldc; k.KT = a.K / BK; k.alpha = a.alpha; k.beta = a.beta;
    k.ntiles = a.ntiles; k.tiles_lower = a.tiles_lower; k.ntj = a.ntj; k.tile_list = a.tile_list;
    k.diag_pad_from = a.diag_pad_from; k.ws = a.ws; k.nwg = a.nwg; k.bk = batch_k(a.batch);
    k.sk_claim = nullptr;
    const int B = a.batch.count;
    if (a.ntiles <= 0 || k.KT <= 0) return hipSuccess;
    if (a.nwg == a.ntiles && !a.s && a.diag_pad_from < 0) {   // one whole tile per workgroup
        if (a.tile_edge == 64 && k.KT == 8)      hipLaunchKernelGGL((gemm_nt_tile_k128_kernel<2, 2>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else if (a.tile_edge == 32 && k.KT == 8) hipLaunchKernelGGL((gemm_nt_tile_k128_kernel<1, 4>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else if (a.tile_edge == 64) hipLaunchKernelGGL((gemm_nt_tile_kernel<2, 2>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else if (a.tile_edge == 32) hipLaunchKernelGGL((gemm_nt_tile_kernel<1, 4>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else                        hipLaunchKernelGGL((gemm_nt_tile_kernel<4, 4>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        return hipGetLastError();
    }
    // a batch of a multiple of 8 LPs: one LP per XCD at a time (see BatchK)
    const bool xm = B >= 8 && B % 8 == 0;
    k.bk.xcd_major = xm ? 1 : 0;
    const dim3 grid = xm ? dim3(8, a.nwg, B / 8) : dim3(a.nwg, 1, B);
    static const bool w4 = getenv("LPIPM_ADAT_W4") != nullptr;   // measurement knob: the 4-wave form (scripts/adat_ab.py)
    static const bool sk_static = getenv("LPIPM_SK_STATIC") != nullptr;   // measurement knob: static stream-K split
    // dynamic claiming of the remainder chunks: single LP, 8-wave kernel, a data-parallel phase exists, chunks divide K
    const int nrem_ = a.ntiles - (a.ntiles / a.nwg) * a.nwg;
    if (a.sk_claim && !w4 && !sk_static && B == 1 && a.ntiles >= a.nwg && nrem_ > 0 && k.KT % SK_CHUNK == 0 && k.KT > SK_CHUNK &&
        nrem_ * (k.KT / SK_CHUNK) <= 2 * a.nwg) {
        hipError_t em = hipMemsetAsync(a.sk_claim, 0, sizeof(unsigned int), st);
        if (em != hipSuccess) return em;
        k.sk_claim = a.sk_claim;
    }
    if (a.s && !w4) hipLaunchKernelGGL(gemm_nt_streamk_w8_kernel<true>, grid, dim3(512), 0, st, k);
    else if (a.s)   hipLaunchKernelGGL(gemm_nt_streamk_kernel<true>, grid, dim3(256), 0, st, k);
    else if (!w4)   hipLaunchKernelGGL(gemm_nt_streamk_w8_kernel<false>, grid, dim3(512), 0, st, k);
    else            hipLaunchKernelGGL(gemm_nt_streamk_kernel<false>, grid, dim3(256), 0, st, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // remainder tiles: a split exists unless every workgroup boundary falls on a tile boundary
    const int nrem = a.ntiles - (a.ntiles / a.nwg) * a.nwg;
    const long long total = (long long)nrem * k.KT;
    bool split = false;
    for (int g = 1; g < a.nwg && !split; ++g) split = ((g * total) / a.nwg) % k.KT != 0;
    if (split || k.sk_claim) {
        hipLaunchKernelGGL(gemm_nt_fixup_kernel, xm ? dim3(8, nrem * FIX_SPLIT, B / 8) : dim3(nrem * FIX_SPLIT, 1, B), dim3(256), 0,
                           st, k);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_gemm_grouped(const GemmTileDesc* descs_dev, int ntiles, hipStream_t st, const Batch& bt, int edge) {
    if (ntiles <= 0) return hipSuccess;
    if (edge == 64) hipLaunchKernelGGL(gemm_nt_grouped64_kernel, dim3(ntiles, 1, bt.count), dim3(256), 0, st, descs_dev, batch_k(bt));
    else            hipLaunchKernelGGL(gemm_nt_grouped_kernel, dim3(ntiles, 1, bt.count), dim3(256), 0, st, descs_dev, batch_k(bt));
    return hipGetLastError();
}

}  // namespace lpipm
