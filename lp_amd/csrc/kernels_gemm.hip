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
//   * data-parallel + stream-K hybrid: whole tiles while they divide over the resident workgroups,
//     the remaining tiles' k-range in CHUNKS claimed through one device word (528 lower tiles at
//     m=4096 do not divide over 512 resident workgroups); a chunk's partial tile goes to a slab and
//     a second pass adds the slabs of a tile in chunk order -- deterministic, no data atomics.
//   * CANONICAL SUMMATION ORDER (A.D.A^T): the contraction is cut into chunks of kc k-tiles (256
//     columns up to n = 4096, the KC panel depth of the reference's dgemm -- matrixmultiply, whose
//     `C += A_panel . B_panel` per packed panel is restated in oracle/oracle_linalg.c); every chunk is
//     summed from zero in k order and the chunk sums are added in chunk order.  A data-parallel tile
//     flushes its accumulators into C at every chunk boundary, a stream-K chunk goes to its slab and
//     the fix-up adds the slabs in the same order: M has the same bits whatever the decomposition
//     (single LP, lockstep batch, any workgroup count), and the rounding error of a length-n sum is
//     that of a two-level sum, like the reference's, instead of a length-n running sum (measured on
//     the 256 C4 members: without it the lockstep path ended one-sidedly further from the vertex).
//   * workgroups are renumbered so that the 64 that share an XCD (and its L2) work on one
//     8x8 super-block of tiles: 16 row panels of A feed 64 tiles.
#include "lpipm_internal.hpp"
#include <cstdlib>

namespace lpipm {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int LDS_STRIDE = BK + 2;  // doubles per LDS row (144 B, keeps 16-B alignment)
// Bank-conflict-free fragment reads.  A ds_read_b128 of a wave is served in four groups of 16 lanes that are NOT lanes
// 16g .. 16g+15 but {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63} (MI355X_MICROARCH.md, LDS
// table): with rows 36 dwords apart and lane (fr, fq) reading the 16-byte k-block fq of row fr, 28 of every 64 lane slots
// collided (PMC, round 3: SQ_LDS_BANK_CONFLICT = 37 % of SQ_LDS_IDX_ACTIVE in the A.D.A^T launch).  Rows 4..11 of every
// 16 keep their k-blocks pairwise swapped (block kb sits at kb ^ 1): every group then covers the 64 banks exactly once
// (exhaustive search over per-row XOR / rotation swizzles; none exists for a plain rotation).  The staging stores of a
// row are its 8 blocks in another order: still one contiguous 128-byte run per 8 lanes.  Which k a lane multiplies is
// unchanged, so are all results.
__device__ __forceinline__ int lds_swz(int row) { return ((row + 4) >> 3) & 1; }
__device__ __forceinline__ int lds_wcol(int row, int scol) { return scol ^ (lds_swz(row) << 1); }     // scol = 2 * k-block
__device__ __forceinline__ int lds_rq(int fr, int fq) { return (fq ^ lds_swz(fr)) << 1; }
constexpr int SK_CHUNK = 16;        // k-tiles per dynamically claimed stream-K chunk when no canonical chunk is set

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
    int kc;                   // summation chunk of a data-parallel tile in k-tiles (0: plain running sum over the whole k-range)
    int sk;                   // stream-K unit in k-tiles (== kc up to KT = 256: one canonical chunking for every tile)
    double* C2;               // nullable: the final value of every tile is stored here too (same ldc)
    BatchK bk;
};
// LP blockIdx.z of a lockstep batch: per-LP pointers shifted (the tile list is shared)
__device__ __forceinline__ GemmK batch_shift(const GemmK& p0) {
    GemmK p = p0;
    p.P = batch_ptr(p0.P, p0.bk); p.Q = batch_ptr(p0.Q, p0.bk); p.s = batch_ptr(p0.s, p0.bk);
    p.C = batch_ptr(p0.C, p0.bk); p.ws = batch_ptr(p0.ws, p0.bk); p.sk_claim = batch_ptr(p0.sk_claim, p0.bk);
    p.C2 = batch_ptr(p0.C2, p0.bk);
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


// acc(128x128 tile, 64 doubles per lane) += sum over k-tiles [kb, ke) of P-panel . (s o Q-panel)^T
// Pp / Qp point at this thread's first staging element (row srow, column scol of the panels).
// MTM x MTN = 16x16 MFMA tiles per wave (2x2 waves per workgroup): workgroup tile = 32*MTM x 32*MTN.
//   4x4 -> 128x128 (throughput launches), 2x2 -> 64x64 and 1x4 -> 32x128 (latency-bound launches)
// PF register stages of global loads are in flight: k-tile k + PF is requested while k-tile k is multiplied, so a
// load has PF - 1 whole k-tile periods to arrive.  With one stage (the first version) the small-tile launches of
// the factorisation -- one round of tiles, nothing else on the CU -- ran at ~1.5 us per k-tile, the latency of an L2 /
// Infinity-Cache load, against 0.43 us of MFMA work (trace: a K = 512 trailing update of 64x64 tiles 50 us).
template <bool SCALE, int MTM, int MTN, int PF = (MTM * MTN >= 16 ? 2 : 3)>
__device__ __forceinline__ void tile_mainloop(double (*ldsA)[32 * MTM][LDS_STRIDE], double (*ldsB)[32 * MTN][LDS_STRIDE],
                                              const double* __restrict__ Pp, long long ldp,
                                              const double* __restrict__ Qp, long long ldq,
                                              const double* __restrict__ s, int kb, int ke, d4 (&acc)[MTM][MTN],
                                              int srow, int scol, int wr, int wc, int fr, int fq) {
    d2 sa[PF][MTM], sb[PF][MTN], sv[PF];
    auto gload = [&](int kt, int st) {
        const long long ko = (long long)kt * BK;
#pragma unroll
        for (int r = 0; r < MTM; ++r) sa[st][r] = *(const d2*)(Pp + (long long)(32 * r) * ldp + ko);
#pragma unroll
        for (int r = 0; r < MTN; ++r) sb[st][r] = *(const d2*)(Qp + (long long)(32 * r) * ldq + ko);
        sv[st] = SCALE ? *(const d2*)(s + ko + scol) : (d2){1.0, 1.0};
    };
    // the scale is applied here, after the MFMAs of the current k-tile: multiplying right after the
    // loads would make the wave wait for the prefetch it has just issued
    auto lstore = [&](int buf, int st) {
#pragma unroll
        for (int r = 0; r < MTM; ++r) *(d2*)&ldsA[buf][srow + 32 * r][lds_wcol(srow, scol)] = sa[st][r];
#pragma unroll
        for (int r = 0; r < MTN; ++r) *(d2*)&ldsB[buf][srow + 32 * r][lds_wcol(srow, scol)] = SCALE ? sb[st][r] * sv[st] : sb[st][r];
    };
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (kb + u < ke) gload(kb + u, u);
    lstore(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt0 = kb; kt0 < ke; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {          // k-tile kt0 + u lives in stage u and is in LDS buffer `cur`
            const int kt = kt0 + u;
            if (kt < ke) {
                if (kt + PF < ke) gload(kt + PF, u);
#pragma unroll
                for (int round = 0; round < 2; ++round) {
                    d2 a[MTM], b[MTN];
#pragma unroll
                    for (int mi = 0; mi < MTM; ++mi)
                        a[mi] = *(const d2*)&ldsA[cur][wr * (16 * MTM) + mi * 16 + fr][round * 8 + lds_rq(fr, fq)];
#pragma unroll
                    for (int nj = 0; nj < MTN; ++nj)
                        b[nj] = *(const d2*)&ldsB[cur][wc * (16 * MTN) + nj * 16 + fr][round * 8 + lds_rq(fr, fq)];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
                            for (int nj = 0; nj < MTN; ++nj)
                                acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);
                }
                if (kt + 1 < ke) lstore(cur ^ 1, (u + 1) % PF);
                __syncthreads();
                cur ^= 1;
            }
        }
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

// ---------------------------------------------------------------------------------------------------------
// A.D.A^T kernel.  128x128 workgroup tile, 512 threads = 2x4 waves of 64x32 (4x2 MFMA tiles, 32 fp64
// accumulators per lane), <= 128 VGPRs, so TWO such workgroups = 4 waves per SIMD are resident per CU.
// Why 8 waves: with 2 waves per SIMD (the first version: 4 waves of 64x64 per workgroup) a wave's non-MFMA phase of a
// k-tile (issue the prefetch, ds_read the fragments, scale + ds_write the next k-tile, barrier: ~4000 cycles under
// contention, s_memtime stamps) is as long as its partner's MFMA phase (64 x 64 cycles), so the two can only just
// cover each other (measured 87 % MFMA-busy); with 4 waves per SIMD each wave issues 32 MFMAs per k-tile and has
// three partners' 6144 cycles of cover (93 %).
//
// chunk_end(q) is called when the k-tile that ends chunk q (kc k-tiles, counted from k-tile 0) has been
// accumulated and more k-tiles follow: the caller flushes and clears the accumulators there, while the software
// pipeline (next k-tile already in LDS) keeps running.
// Addressing: buffer loads.  A tile's row panel is one buffer resource (wave-uniform origin in SGPRs), the k advance
// and the +64-row step go into the scalar offset, and a lane contributes ONE 32-bit byte offset (its staging row and
// column): the five loads of a k-tile cost one VGPR of addresses instead of ten.  At 128 VGPRs per wave (4 waves per
// SIMD) every register the compiler spills inside this loop puts an `s_waitcnt vmcnt(0)` in front of the prefetch it
// has just issued.  (A panel of 128 rows must stay below 4 GiB: ld < 4M columns, checked at launch.)
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ d2 buf_load_d2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ double buf_load_d(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store_d(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, v), r, (int)voff, (int)soff, 0);
}
// the same with a cache-policy operand (16 = sc1: write-through / L1-bypassing, see the units kernel).  The value goes
// through a by-value double: a bit_cast written directly on `acc[mi][nj][r]` inside the unrolled loops stored element 0
// of the accumulator quad for every r (hipcc 7.2, seen on the GPU: rows fq + 4r, r > 0, received row fq's values).
template <int AUX>
__device__ __forceinline__ void buf_store_d_aux(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, v), r, (int)voff, (int)soff, AUX);
}
template <int AUX>
__device__ __forceinline__ void buf_store_d2_aux(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, d2 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, (int)voff, (int)soff, AUX);
}
template <int AUX>
__device__ __forceinline__ d2 buf_load_d2_aux(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}

// Diagonal tiles of a symmetric product (A.D.A^T): wave (wr, wc) of the 2 x 4 layout owns the 16x16 blocks with block row
// 4 wr + mi and block column 2 wc + nj of the tile; a block is strictly above the diagonal when its column index exceeds
// its row index, i.e. d + nj > mi with d = 2 wc - 4 wr.  Patterns: 0 nothing skipped (d <= -2), 1 (d = 0), 2 (d = 2),
// 3 everything (d >= 4).  28 of a diagonal tile's 64 blocks go: 3 % of the MFMA work at 4096x8192, 11 % at 1024x2048.
__device__ __forceinline__ int diag_pattern(int wr, int wc) {
    const int d = 2 * wc - 4 * wr;
    return d < 0 ? 0 : (d == 0 ? 1 : (d == 2 ? 2 : 3));
}
// Which 64 x 32 part of the tile wave w takes.  Waves w and w + 4 share a SIMD (a 512-thread workgroup's waves go round the
// four SIMDs), and a workgroup advances at the pace of its busiest SIMD (one barrier per k-tile), so on a diagonal tile the
// blocks left (7, 3, 0, 0 / 8, 8, 7, 3 of 8 for wc = 0..3 in rows wr = 0 / 1) are paired to 8, 8, 10, 10 per SIMD.
__device__ __forceinline__ void wave_part(int wave, int& wr, int& wc) {
    wr = (0x4B >> wave) & 1;             // w: 0 1 2 3 4 5 6 7 -> wr 1 1 0 1 0 0 1 0
    wc = (0x7E84 >> (2 * wave)) & 3;     //                       wc 0 1 0 2 2 3 3 1
}
// the first pattern that leaves block (mi, nj) out: the block is issued while the wave's pattern is below it
__device__ __forceinline__ constexpr int diag_group(int mi, int nj) {
    return nj > mi ? 1 : (2 + nj > mi ? 2 : 3);
}

// f.template operator()<PAT>() for the wave-uniform pattern `pat`
template <typename F> __device__ __forceinline__ void diag_dispatch(int pat, F&& f) {
    if (pat == 0) f.template operator()<0>();
    else if (pat == 1) f.template operator()<1>();
    else if (pat == 2) f.template operator()<2>();
    else f.template operator()<3>();
}

// One pass over the k-tiles [kb, ke) of a tile, in chunks that end at multiples of kc k-tiles (kc == 0: one chunk).
// The software pipeline (next k-tile prefetched into registers while the current one is multiplied out of LDS) runs
// across chunk boundaries; the hot inner loop is the plain k-tile loop and the chunk logic lives around it:
//   touch(first)  at the start of a chunk's last k-tile: may issue loads that pull the C tile towards L2
//   flush(first, last)  after a chunk's last k-tile: stores / adds the accumulators (the caller's business); the
//                 accumulators restart from zero if more chunks follow.
template <bool SCALE, int PAT = 0, typename FL, typename TC>
__device__ __forceinline__ void tile_pass_w8(double (*ldsA)[TILE][LDS_STRIDE], double (*ldsB)[TILE][LDS_STRIDE],
                                             __amdgpu_buffer_rsrc_t Pr, unsigned p64, __amdgpu_buffer_rsrc_t Qr, unsigned q64,
                                             __amdgpu_buffer_rsrc_t Sr, unsigned offP, unsigned offQ, unsigned offS,
                                             int kb, int ke, d4 (&acc)[4][2],
                                             int srow, int scol, int wr, int wc, int fr, int fq, int kc, FL&& flush,
                                             TC&& touch) {
    d2 sa[2], sb[2], sv = (d2){1.0, 1.0};
    auto gload = [&](int kt) {
        const unsigned ko = (unsigned)kt * (unsigned)(BK * sizeof(double));
        if (SCALE) sv = buf_load_d2(Sr, offS, ko);
        sa[0] = buf_load_d2(Pr, offP, ko);
        sa[1] = buf_load_d2(Pr, offP, ko + p64);
        sb[0] = buf_load_d2(Qr, offQ, ko);
        sb[1] = buf_load_d2(Qr, offQ, ko + q64);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 2; ++r) *(d2*)&ldsA[buf][srow + 64 * r][lds_wcol(srow, scol)] = sa[r];
#pragma unroll
        for (int r = 0; r < 2; ++r) *(d2*)&ldsB[buf][srow + 64 * r][lds_wcol(srow, scol)] = SCALE ? sb[r] * sv : sb[r];
    };
    // PAT > 0: the pass of a wave over a DIAGONAL tile of a symmetric product whose blocks of groups <= PAT (diag_group) lie
    // strictly above the diagonal and are left out -- nothing reads them.  A straight-line body per pattern: the caller
    // switches on the wave's pattern once per pass (diag_dispatch), every other tile runs PAT = 0.
    auto mfma_ktile = [&](int cur) {
        if constexpr (PAT < 3) {
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                d2 a[4] = {}, b[2] = {};
                if (round == 0) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    if (diag_group(mi, 0) > PAT) a[mi] = *(const d2*)&ldsA[cur][wr * 64 + mi * 16 + fr][round * 8 + lds_rq(fr, fq)];
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
                    if (diag_group(3, nj) > PAT) b[nj] = *(const d2*)&ldsB[cur][wc * 32 + nj * 16 + fr][round * 8 + lds_rq(fr, fq)];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int nj = 0; nj < 2; ++nj)
                            if (diag_group(mi, nj) > PAT)
                                acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(2);   // the short non-MFMA phase goes first: it is what the partners wait for
    };
    gload(kb);
    lstore(0);
    __syncthreads();
    int cur = 0, kt = kb;
    bool first = true;
    while (kt < ke) {
        int ce = ke;                                       // end of this chunk
        if (kc > 0) { const int e = (kt / kc + 1) * kc; ce = e < ke ? e : ke; }
        for (; kt < ce - 1; ++kt) {                        // the hot loop
            gload(kt + 1);
            mfma_ktile(cur);
            lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
        const bool more = ce < ke;                         // last k-tile of the chunk
        unsigned pf0 = 0, pf1 = 0;
        touch(first, pf0, pf1);                            // (before the prefetch: whatever its addresses need reloaded must
        if (more) gload(ce);                               //  not wait behind the loads issued here)
        mfma_ktile(cur);
        if (more) lstore(cur ^ 1);
        asm volatile("" :: "v"(pf0), "v"(pf1));            // the touch loads have landed (and their registers are free) from here
        flush(first, !more);
        if (more) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        first = false;
        __syncthreads();
        cur ^= 1;
        kt = ce;
    }
    __builtin_amdgcn_s_setprio(0);
}

// Workgroup g of an LP:
//   phase 1 (data-parallel): tiles g, g + nwg, ... of the first ntiles_dp = floor(ntiles/nwg)*nwg tiles, whole k-range
//            each.  Every workgroup walks k from 0 in lockstep, so the ~64 workgroups of an XCD (one 8x8 super-block
//            of tiles) hit each other's A panels in that XCD's L2.  With a canonical chunk (p.kc) the accumulators
//            are added into the C tile at every chunk boundary and restart from zero.
//   phase 2 (stream-K): the remaining tiles' k-range in chunks (p.kc, or SK_CHUNK k-tiles), claimed through one
//            device word per LP: the data-parallel tiles do not all finish together (stamps at C3: 1876 .. 2025 us
//            after launch), so chunks go to whoever is free.  Slab c holds chunk c; gemm_nt_fixup_kernel adds a
//            tile's slabs in chunk order -- the same sums in the same order as phase 1's flushes.
template <bool SCALE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_streamk_w8_kernel(const GemmK p0) {
    if (batch_done(p0.bk)) return;
    const GemmK p = batch_shift(p0);
    __shared__ __attribute__((aligned(16))) double ldsA[2][TILE][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][TILE][LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wr, wc;                                       // 2 x 4 waves: 64 rows x 32 columns each
    wave_part(wave, wr, wc);
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
    // C tile (ti, tj) as a buffer; this lane's element of MFMA block (mi, nj), register r sits at
    //   offC + ((mi*16 + 4r)*ldc + nj*16) * 8   (wave-uniform second term)
    const unsigned rowC = (unsigned)(p.ldc * (long long)sizeof(double));
    const unsigned offC = (unsigned)(wr * 64 + fq) * rowC + (unsigned)((wc * 32 + fr) * sizeof(double));
    auto c_rsrc = [&](int ti, int tj) { return make_rsrc(p.C + (long long)(ti * TILE) * p.ldc + tj * TILE, (unsigned)TILE * rowC); };
    auto store_tile = [&](const d4 (&acc)[4][2], int ti, int tj) {
        double* cb = p.C + (long long)(ti * TILE + wr * 64 + fq) * p.ldc + (tj * TILE + wc * 32 + fr);
        tile_store<4, 2>(cb, p.ldc, acc, p.alpha, p.beta, p.diag_pad_from >= 0 && ti == tj, ti * TILE + wr * 64 + fq,
                         p.diag_pad_from, fr, fq, wc * 32 - wr * 64);
    };
    // C tile += alpha * chunk sum.  Two rounds of 16 values per lane: the registers of
    // the staging and fragment values, dead at this point, hold the C values on their way in.
    auto add_tile = [&](const d4 (&acc)[4][2], int ti, int tj, bool final_copy) {
        const __amdgpu_buffer_rsrc_t cr = c_rsrc(ti, tj);
        const __amdgpu_buffer_rsrc_t cr2 = make_rsrc((final_copy ? p.C2 : p.C) + (long long)(ti * TILE) * p.ldc + tj * TILE, (unsigned)TILE * rowC);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double cv[2][4][2];
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
                        cv[m2][r][nj] = buf_load_d(cr, offC, (unsigned)((2 * h + m2) * 16 + 4 * r) * rowC + nj * 128);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
                    {
                        const double v = p.alpha == 1.0 ? cv[m2][r][nj] + acc[2 * h + m2][nj][r] : fma(p.alpha, acc[2 * h + m2][nj][r], cv[m2][r][nj]);
                        buf_store_d(cr, offC, (unsigned)((2 * h + m2) * 16 + 4 * r) * rowC + nj * 128, v);
                        if (final_copy) buf_store_d(cr2, offC, (unsigned)((2 * h + m2) * 16 + 4 * r) * rowC + nj * 128, v);
                    }
        }
    };
    // Touches every 128-B line of this wave's 64 x 32 part of the C tile (lane l: row l, columns 0 and 16), one k-tile
    // before add_tile reads it: the tile was written a chunk ago and has left the L2 since (A streams through it), so
    // the flush would otherwise pay two memory round trips (~6.5 us per flush measured, ~1.5 with the lines in L2).
    auto touch_tile = [&](int ti, int tj, unsigned& pf0, unsigned& pf1) {
        const __amdgpu_buffer_rsrc_t cr = c_rsrc(ti, tj);
        const unsigned off = (unsigned)(wr * 64 + lane) * rowC + (unsigned)(wc * 32 * sizeof(double));
        pf0 = __builtin_amdgcn_raw_buffer_load_b32(cr, (int)off, 0, 0);
        pf1 = __builtin_amdgcn_raw_buffer_load_b32(cr, (int)off, 128, 0);
    };
    const unsigned rowP = (unsigned)(p.ldp * (long long)sizeof(double)), rowQ = (unsigned)(p.ldq * (long long)sizeof(double));
    const unsigned offP = (unsigned)srow * rowP + (unsigned)(scol * sizeof(double));
    const unsigned offQ = SCALE ? offP : (unsigned)srow * rowQ + (unsigned)(scol * sizeof(double));   // A.D.A^T: P = Q = A
    const unsigned offS = (unsigned)(scol * sizeof(double));
    const unsigned p64 = 64u * rowP, q64 = 64u * rowQ;
    auto p_rsrc = [&](int ti) { return make_rsrc(p.P + (long long)(ti * TILE) * p.ldp, (unsigned)TILE * rowP); };
    auto q_rsrc = [&](int tj) { return make_rsrc(p.Q + (long long)(tj * TILE) * p.ldq, (unsigned)TILE * rowQ); };
    const __amdgpu_buffer_rsrc_t Sr = make_rsrc(SCALE ? p.s : p.P, (unsigned)KT * (unsigned)(BK * sizeof(double)));
    for (int tile = g; tile < ntiles_dp; tile += p.nwg) {
        int ti, tj;
        tile_coords(p, tile, ti, tj);
        d4 acc[4][2];
        zero(acc);
        tile_pass_w8<SCALE>(ldsA, ldsB, p_rsrc(ti), p64, q_rsrc(tj), q64, Sr, offP, offQ, offS, 0, KT, acc, srow, scol, wr, wc, fr, fq,
                            p.kc, [&](bool first, bool last) {
                                if (first) store_tile(acc, ti, tj); else add_tile(acc, ti, tj, last && p.C2 != nullptr);
                            },
                            [&](bool first, unsigned& pf0, unsigned& pf1) { if (!first) touch_tile(ti, tj, pf0, pf1); });
    }
    if (ntiles_dp == p.ntiles) return;
    __shared__ int s_claim;
    const int ch_tiles = p.sk;
    const int cpt = (KT + ch_tiles - 1) / ch_tiles;          // stream-K units per tile
    const int nchunks = (p.ntiles - ntiles_dp) * cpt;
    for (;;) {
        if (tid == 0) s_claim = (int)atomicAdd(p.sk_claim, 1u);
        __syncthreads();
        const int ch = __builtin_amdgcn_readfirstlane(s_claim);
        __syncthreads();
        if (ch >= nchunks) break;
        // chunk-major order: the units in flight together are the same k-range of different tiles, which share the
        // row panels of A in L2 (tile-major order would have every unit load two panels of its own)
        const int nrem = p.ntiles - ntiles_dp;
        const int q = ch / nrem, rt = ch - q * nrem;
        const int kb = q * ch_tiles, ke = kb + ch_tiles < KT ? kb + ch_tiles : KT;
        int ti, tj;
        tile_coords(p, ntiles_dp + rt, ti, tj);
        d4 acc[4][2];
        zero(acc);
        tile_pass_w8<SCALE>(ldsA, ldsB, p_rsrc(ti), p64, q_rsrc(tj), q64, Sr, offP, offQ, offS, kb, ke, acc, srow, scol, wr, wc, fr, fq,
                            0, [&](bool, bool) {
            if (cpt == 1) { store_tile(acc, ti, tj); return; }   // the chunk is the whole tile
            const __amdgpu_buffer_rsrc_t wr_ = make_rsrc(p.ws + ((long long)rt * cpt + q) * (TILE * TILE), (unsigned)(TILE * TILE * sizeof(double)));
            const unsigned offW = (unsigned)(((wr * 64 + fq) * TILE + wc * 32 + fr) * sizeof(double));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
                        buf_store_d(wr_, offW, (unsigned)(((mi * 16 + 4 * r) * TILE + nj * 16) * sizeof(double)), acc[mi][nj][r]);
        }, [](bool, unsigned&, unsigned&) {});
    }
}

// ---------------------------------------------------------------------------------------------------------
// A.D.A^T as (tile, chunk) units with an in-launch combine (AdatUnitsArgs, lpipm_internal.hpp).
// Hand-off protocol (MI355X_MICROARCH.md, Workgroup dispatch / inter-workgroup visibility; cdna_hip_programming.md
// Guideline 16 in its counter form): per-XCD L2s are not coherent, so
//   producer : slab stores are WRITE-THROUGH (sc1) -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier ->
//              ONE lane adds to the tile's counter (relaxed, agent scope);
//   consumer : the workgroup whose add completed the tile (told by the value the add returned) -> that lane's
//              agent-scope acquire + s_waitcnt vmcnt(0) -> workgroup barrier -> EVERY load of the slabs is an sc1 load.
// The finished tile of M is stored write-through as well when group words are signalled: its readers are other
// kernels (the factorisation's chain, on another stream) that start while this launch is still running -- their
// kernel-start acquire drops stale lines, but nothing would write this XCD's dirty lines back before this launch ends.
constexpr int AUX_SC1 = 16;
struct UnitsK {
    const double* A; long long lda;
    const double* s;
    double* C; long long ldc;
    double* C2;
    int KT, kc, cpt;
    int nbig, ks;             // chunk q covers k-tiles [q*kc, (q+1)*kc) for q < nbig, then pieces of ks k-tiles (adat_units_chunking)
    int ntiles;
    const int2* tile_list;
    const int2* unit_list;
    int nunits, upc;
    int diag_pad_from;
    double* slabs;
    unsigned int* tile_cnt;
    unsigned int* grp_cnt;
    int grp_w;
    BatchK bk;
};
template <bool GRP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_units_kernel(const UnitsK p0) {
    if (batch_done(p0.bk)) return;
    UnitsK p = p0;
    p.A = batch_ptr(p0.A, p0.bk); p.s = batch_ptr(p0.s, p0.bk); p.C = batch_ptr(p0.C, p0.bk); p.C2 = batch_ptr(p0.C2, p0.bk);
    p.slabs = batch_ptr(p0.slabs, p0.bk); p.tile_cnt = batch_ptr(p0.tile_cnt, p0.bk); p.grp_cnt = batch_ptr(p0.grp_cnt, p0.bk);
    __shared__ __attribute__((aligned(16))) double ldsA[2][TILE][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][TILE][LDS_STRIDE];
    __shared__ unsigned int s_old;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wr, wc;                                       // 2 x 4 waves: 64 rows x 32 columns each
    wave_part(wave, wr, wc);
    const int fr = lane & 15, fq = lane >> 4;
    const int srow = tid >> 3, scol = (tid & 7) * 2;  // staging: 64 rows per pass
    const int b = p.bk.xcd_major ? (int)blockIdx.y : (int)blockIdx.x;
    const int2 un = p.unit_list[b];          // the list is already dealt to the XCDs (solver.hip, deal_units)
    const int tile = un.x, q0 = un.y;
    if (tile < 0) return;                    // padding of an XCD's shorter list
    const int q1 = q0 + p.upc < p.cpt ? q0 + p.upc : p.cpt;
    const int2 tc = p.tile_list[tile];
    const int ti = tc.x, tj = tc.y;
    const int KT = p.KT;
    const unsigned rowA = (unsigned)(p.lda * (long long)sizeof(double));
    const unsigned offP = (unsigned)srow * rowA + (unsigned)(scol * sizeof(double));
    const unsigned offS = (unsigned)(scol * sizeof(double));
    const unsigned p64 = 64u * rowA;
    const __amdgpu_buffer_rsrc_t Pr = make_rsrc(p.A + (long long)(ti * TILE) * p.lda, (unsigned)TILE * rowA);
    const __amdgpu_buffer_rsrc_t Qr = make_rsrc(p.A + (long long)(tj * TILE) * p.lda, (unsigned)TILE * rowA);
    const __amdgpu_buffer_rsrc_t Sr = make_rsrc(p.s, (unsigned)KT * (unsigned)(BK * sizeof(double)));
    // this tile's slabs as ONE buffer: slab q at byte q * 128 KiB
    constexpr unsigned SLAB_BYTES = (unsigned)(TILE * TILE * sizeof(double));
    const __amdgpu_buffer_rsrc_t Wr = make_rsrc(p.slabs + (long long)tile * p.cpt * (TILE * TILE), (unsigned)p.cpt * SLAB_BYTES);
    d4 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};
    // chunk boundaries: nbig chunks of kc k-tiles, the rest of the contraction in pieces of ks
    auto chunk_begin = [&](int q) { const int b0 = q <= p.nbig ? q * p.kc : p.nbig * p.kc + (q - p.nbig) * p.ks; return b0 < KT ? b0 : KT; };
    const int kb = chunk_begin(q0), ke = chunk_begin(q1);
    const int dpat = ti == tj ? diag_pattern(wr, wc) : 0;
    if (p.cpt == 1) {
        // a contraction of one chunk: the unit is the whole tile, stored directly
        diag_dispatch(dpat, [&]<int PAT>() {
            tile_pass_w8<true, PAT>(ldsA, ldsB, Pr, p64, Qr, p64, Sr, offP, offP, offS, 0, KT, acc, srow, scol, wr, wc, fr, fq, 0,
                                    [&](bool, bool) {}, [](bool, unsigned&, unsigned&) {});
        });
        const unsigned rowC = (unsigned)(p.ldc * (long long)sizeof(double));
        const unsigned offC = (unsigned)(wr * 64 + fq) * rowC + (unsigned)((wc * 32 + fr) * sizeof(double));
        const __amdgpu_buffer_rsrc_t cr = make_rsrc(p.C + (long long)(ti * TILE) * p.ldc + tj * TILE, (unsigned)TILE * rowC);
        const __amdgpu_buffer_rsrc_t cr2 = make_rsrc((p.C2 ? p.C2 : p.C) + (long long)(ti * TILE) * p.ldc + tj * TILE, (unsigned)TILE * rowC);
        const bool pad = p.diag_pad_from >= 0 && ti == tj;
        const int row0 = ti * TILE + wr * 64 + fq, dd = wc * 32 - wr * 64;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) {
                    double v = acc[mi][nj][r];
                    if (pad && (mi - nj) * 16 + fq + 4 * r - fr == dd && row0 + mi * 16 + 4 * r >= p.diag_pad_from) v = 1.0;
                    const unsigned so = (unsigned)(mi * 16 + 4 * r) * rowC + nj * 128;
                    buf_store_d_aux<GRP ? AUX_SC1 : 0>(cr, offC, so, v);
                    if (p.C2) buf_store_d_aux<0>(cr2, offC, so, v);
                }
    } else {
        int q = q0;
        const unsigned offW = (unsigned)(((wr * 64 + fq) * TILE + wc * 32 + fr) * sizeof(double));
        diag_dispatch(dpat, [&]<int PAT>() {
        tile_pass_w8<true, PAT>(ldsA, ldsB, Pr, p64, Qr, p64, Sr, offP, offP, offS, kb, ke, acc, srow, scol, wr, wc, fr, fq, p.kc,
                           [&](bool, bool) {
                               const unsigned sb = (unsigned)q * SLAB_BYTES;
#pragma unroll
                               for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                                   for (int r = 0; r < 4; ++r)
#pragma unroll
                                       for (int nj = 0; nj < 2; ++nj)
                                           buf_store_d_aux<AUX_SC1>(Wr, offW, sb + (unsigned)(((mi * 16 + 4 * r) * TILE + nj * 16) * sizeof(double)),
                                                                    acc[mi][nj][r]);
                               ++q;
                           },
                           [](bool, unsigned&, unsigned&) {});
        });
        // publish: every storing wave drains its stores, then ONE lane adds this unit's chunks to the tile's counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_old = __hip_atomic_fetch_add(p.tile_cnt + tile, (unsigned)(q1 - q0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if ((int)s_old + (q1 - q0) != p.cpt) return;           // not the last arriver of this tile (workgroup-uniform)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // every arrival of this launch is in: the word goes back to zero for the next launch (the host then needs no
            // memset per launch; with group words the consumers on other streams make the host clear both kinds)
            if (!GRP) __hip_atomic_store(p.tile_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        // combine: slabs 0 .. cpt-1 added in chunk order; a thread owns 16 pairs of adjacent elements, 1024 elements apart
        const unsigned rowC = (unsigned)(p.ldc * (long long)sizeof(double));
        const __amdgpu_buffer_rsrc_t cr = make_rsrc(p.C + (long long)(ti * TILE) * p.ldc + tj * TILE, (unsigned)TILE * rowC);
        const __amdgpu_buffer_rsrc_t cr2 = make_rsrc((p.C2 ? p.C2 : p.C) + (long long)(ti * TILE) * p.ldc + tj * TILE, (unsigned)TILE * rowC);
        const unsigned voff = (unsigned)tid * 16u;
        const int cpt = p.cpt;
        for (int i = 0; i < 16; i += 2) {
            d2 sum[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                sum[a] = buf_load_d2_aux<AUX_SC1>(Wr, voff, (unsigned)(i + a) * 8192u);
            for (int qb = 1; qb < cpt; qb += 4) {
                d2 v[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        if (qb + j < cpt)
                            v[a][j] = buf_load_d2_aux<AUX_SC1>(Wr, voff, (unsigned)(qb + j) * SLAB_BYTES + (unsigned)(i + a) * 8192u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        if (qb + j < cpt) sum[a] += v[a][j];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int e = tid * 2 + (i + a) * 1024;
                const int r = e >> 7, c = e & 127;
                const int row = ti * TILE + r, col = tj * TILE + c;
                d2 v = sum[a];
                if (p.diag_pad_from >= 0 && ti == tj) {
                    if (row == col && row >= p.diag_pad_from) v[0] = 1.0;
                    if (row == col + 1 && row >= p.diag_pad_from) v[1] = 1.0;
                }
                const unsigned co = (unsigned)r * rowC + (unsigned)(c * sizeof(double));
                buf_store_d2_aux<GRP ? AUX_SC1 : 0>(cr, co, 0u, v);
                if (p.C2) buf_store_d2_aux<0>(cr2, co, 0u, v);
            }
        }
    }
    if (GRP) {   // this tile of M is complete: drain its (write-through) stores, then count it in its column group's word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) (void)__hip_atomic_fetch_add(p.grp_cnt + tj / p.grp_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The device-side wait of a stream: see launch_wait_count.
__global__ __launch_bounds__(64) void wait_count_kernel(const unsigned int* cnt, unsigned int target, const int* done,
                                                        unsigned int* timeout) {
    if (threadIdx.x != 0) return;
    if (done && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // the producer returned at once
    for (unsigned int spins = 0; spins < (1u << 21); ++spins) {        // ~1 us per poll: gives up after a few seconds
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
        __builtin_amdgcn_s_sleep(16);
    }
    __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        for (int r = 0; r < MTM; ++r) *(d2*)&ldsA[buf][srow + 32 * r][lds_wcol(srow, scol)] = pa[kt][r];
#pragma unroll
        for (int r = 0; r < MTN; ++r) *(d2*)&ldsB[buf][srow + 32 * r][lds_wcol(srow, scol)] = pb[kt][r];
        __syncthreads();   // buffer `buf` was last read two k-tiles ago: that read finished before the previous barrier
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            d2 a[MTM], b[MTN];
#pragma unroll
            for (int mi = 0; mi < MTM; ++mi)
                a[mi] = *(const d2*)&ldsA[buf][wr * (16 * MTM) + mi * 16 + fr][round * 8 + lds_rq(fr, fq)];
#pragma unroll
            for (int nj = 0; nj < MTN; ++nj)
                b[nj] = *(const d2*)&ldsB[buf][wc * (16 * MTN) + nj * 16 + fr][round * 8 + lds_rq(fr, fq)];
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

// The in-place panel solve of the factorisation (L21 = A21 . inv(L_kk)^T): a workgroup must own whole rows, so its tile is
// 32 rows x all 128 columns; on 8 waves (2 x 4 waves of 16 x 32) a wave runs 64 MFMAs instead of the 128 of the 4-wave
// <1, 4> tile -- the launch is one round of tiles on a chain, its length is the per-wave MFMA count plus one round trip
// (factorisation at m = 4096: 1909 -> 1896 us).
__global__ __launch_bounds__(512, 1) void gemm_nt_rows32_k128_kernel(const GemmK p0) {
    if (batch_done(p0.bk)) return;
    const GemmK p = batch_shift(p0);
    constexpr int TM = 32, TN = 128, KT8 = 8;
    __shared__ __attribute__((aligned(16))) double ldsA[2][TM][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][TN][LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int srow = tid >> 3, scol = (tid & 7) * 2;          // staging: 64 rows per pass, 8 lanes per 128-B row segment
    int ti, tj;
    tile_coords(p, blockIdx.x, ti, tj);
    const bool stage_a = srow < TM;
    const double* Pp = p.P + (long long)(ti * TM + (stage_a ? srow : 0)) * p.ldp + scol;
    const double* Qp = p.Q + (long long)(tj * TN + srow) * p.ldq + scol;
    d2 pa[KT8], pb[KT8][2];
#pragma unroll
    for (int kt = 0; kt < KT8; ++kt) {
        pa[kt] = *(const d2*)(Pp + kt * BK);
#pragma unroll
        for (int r = 0; r < 2; ++r) pb[kt][r] = *(const d2*)(Qp + (long long)(64 * r) * p.ldq + kt * BK);
    }
    d4 acc[1][2];
    acc[0][0] = acc[0][1] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < KT8; ++kt) {
        const int buf = kt & 1;
        if (stage_a) *(d2*)&ldsA[buf][srow][lds_wcol(srow, scol)] = pa[kt];
#pragma unroll
        for (int r = 0; r < 2; ++r) *(d2*)&ldsB[buf][srow + 64 * r][lds_wcol(srow, scol)] = pb[kt][r];
        __syncthreads();   // buffer `buf` was last read two k-tiles ago: that read finished before the previous barrier
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const d2 a = *(const d2*)&ldsA[buf][wr * 16 + fr][round * 8 + lds_rq(fr, fq)];
            d2 b[2];
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) b[nj] = *(const d2*)&ldsB[buf][wc * 32 + nj * 16 + fr][round * 8 + lds_rq(fr, fq)];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
                    acc[0][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[nj][t], acc[0][nj], 0, 0, 0);
        }
    }
    double* cb = p.C + (long long)(ti * TM + wr * 16 + fq) * p.ldc + (tj * TN + wc * 32 + fr);
    tile_store<1, 2>(cb, p.ldc, acc, p.alpha, p.beta, false, 0, -1, fr, fq);
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

// And with 32x32 output tiles: the merge stages of one LP are one round of tiles each, so a stage lasts as long as its longest
// k-loop, and a quarter-size tile has the shortest MFMA chain and the tightest triangular k-range.
__global__ __launch_bounds__(256, 4) void gemm_nt_grouped32_kernel(const GemmTileDesc* __restrict__ descs, BatchK bk) {
    if (batch_done(bk)) return;
    __shared__ __attribute__((aligned(16))) double ldsA[2][32][LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) double ldsB[2][32][LDS_STRIDE];
    TILE_THREAD_IDS
    GemmTileDesc d = descs[blockIdx.x];
    d.P = batch_ptr(d.P, bk); d.Q = batch_ptr(d.Q, bk); d.C = batch_ptr(d.C, bk);
    d4 acc[1][1];
    acc[0][0] = (d4){0.0, 0.0, 0.0, 0.0};
    tile_mainloop<false, 1, 1>(ldsA, ldsB, d.P + (long long)srow * d.ldp + scol, d.ldp, d.Q + (long long)srow * d.ldq + scol,
                               d.ldq, nullptr, d.kt_begin, d.kt_end, acc, srow, scol, wr, wc, fr, fq);
    double* cb = d.C + (long long)(wr * 16 + fq) * d.ldc + (wc * 16 + fr);
    tile_store<1, 1>(cb, d.ldc, acc, d.alpha, 0.0, false, 0, -1, fr, fq);
}

// Adds the chunk slabs of every stream-K (remainder) tile in chunk order.  grid = remainder tiles x FIX_SPLIT:
// a tile can have dozens of slabs, so its 16K elements are spread over FIX_SPLIT workgroups (8 rows each) to
// keep this pass off the critical path.
constexpr int FIX_SPLIT = 16;
__global__ __launch_bounds__(256) void gemm_nt_fixup_kernel(const GemmK p0) {
    if (batch_done(p0.bk)) return;
    const GemmK p = batch_shift(p0);
    const int ntiles_dp = (p.ntiles / p.nwg) * p.nwg;
    const int bx = p.bk.xcd_major ? (int)blockIdx.y : (int)blockIdx.x;
    const int rt = bx / FIX_SPLIT, part = bx % FIX_SPLIT;
    const int cpt = (p.KT + p.sk - 1) / p.sk;                // slabs [rt*cpt, (rt+1)*cpt), one per stream-K unit
    const int q0 = 0, q1 = cpt;
    int ti, tj;
    tile_coords(p, ntiles_dp + rt, ti, tj);
    constexpr int PER = TILE * TILE / FIX_SPLIT;
    const double* slab0 = p.ws + (long long)rt * cpt * (TILE * TILE);
    for (int e = part * PER + threadIdx.x * 2; e < (part + 1) * PER; e += 512) {
        d2 sum = *(const d2*)(slab0 + (long long)q0 * (TILE * TILE) + e);
        for (int q = q0 + 1; q < q1; ++q) sum += *(const d2*)(slab0 + (long long)q * (TILE * TILE) + e);
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
        if (p.C2) *(d2*)(p.C2 + (long long)row * p.ldc + col) = v;
    }
}

// Canonical chunk of the A.D.A^T contraction in k-tiles: 128 columns up to n = 1024 (a small LP has few tiles: more,
// shorter stream-K units fill more CUs), 256 columns (the reference dgemm's KC) up to n = 4096, 1024 columns above.
// A data-parallel tile's flush is a read-modify-write of its C tile (~10 us, and 256 KB of fabric traffic); measured at
// C3 on one box, per launch and distance of the solve's x from the planted vertex:
//   one running sum 5.5e-7 | 1024 columns 2.27 ms, 1.2e-7 | 512 columns 2.34 ms, 7.9e-8 | 256 columns 2.55 ms, 6.3e-8
// (the oracle itself: 6.3e-8).  LPIPM_ADAT_KC=<k-tiles> overrides the rule (measurement knob).
int gemm_streamk_chunk(int KT) {
    static const int forced = lp_knob("LPIPM_ADAT_KC") ? atoi(lp_knob("LPIPM_ADAT_KC")) : -1;
    if (forced >= 0) return forced == 0 ? (KT > 0 ? KT : 1) : forced;
    return KT <= 64 ? 8 : (KT <= 256 ? 16 : 64);
}
// Stream-K unit for a contraction of KT k-tiles: the canonical chunk up to KT = 256 (ONE chunking for data-parallel and
// stream-K tiles: M's bits do not depend on the decomposition -- the sizes that run both alone and as lockstep batches);
// above, 16 k-tiles whatever the data-parallel chunk: the launch ends one unit after the ideal time at best, and a unit
// of 64 k-tiles is 0.24 ms (C3: the 16 stream-K tiles cost 0.25 ms for 3 % of the work).  Those tiles are then summed in
// finer blocks than the data-parallel ones (same determinism, the decomposition-independence is given up for big LPs).
static int streamk_unit(int KT) {
    const int kc = gemm_streamk_chunk(KT);
    static const int forced = lp_knob("LPIPM_ADAT_SK") ? atoi(lp_knob("LPIPM_ADAT_SK")) : 0;   // measurement knob
    if (forced > 0 && KT > 256) return forced;
    return KT <= 256 ? kc : (kc < 16 ? kc : 16);
}
// stream-K units per tile (1: no split, the tile is one running sum)
static int streamk_cpt(int KT) {
    if (KT <= gemm_streamk_chunk(KT)) return 1;
    const int u = streamk_unit(KT);
    return (KT + u - 1) / u;
}
int gemm_streamk_nwg(int ntiles, int KT, int num_cu) {
    const int cpt = streamk_cpt(KT);
    const long long units = (long long)ntiles * cpt;   // (tile, unit) work items
    long long nwg = 2LL * num_cu;              // 2 resident workgroups per CU
    if (nwg > units) nwg = units;
    if (nwg < 1) nwg = 1;
    return (int)nwg;
}
bool gemm_streamk_split(int KT) { return streamk_cpt(KT) > 1; }
size_t gemm_streamk_slabs(int ntiles, int KT, int nwg) {
    const int cpt = streamk_cpt(KT);
    const int nrem = ntiles - (ntiles / nwg) * nwg;
    return cpt > 1 ? (size_t)nrem * cpt : 0;
}

hipError_t launch_gemm_nt(const GemmArgs& a, hipStream_t st) {
    GemmK k;
    k.P = a.P; k.ldp = a.ldp; k.Q = a.Q; k.ldq = a.ldq; k.s = a.s;
    k.C = a.C; k.ldc = a.ldc; k.KT = a.K / BK; k.alpha = a.alpha; k.beta = a.beta;
    k.ntiles = a.ntiles; k.tiles_lower = a.tiles_lower; k.ntj = a.ntj; k.tile_list = a.tile_list;
    k.diag_pad_from = a.diag_pad_from; k.ws = a.ws; k.nwg = a.nwg; k.bk = batch_k(a.batch);
    k.sk_claim = a.sk_claim; k.kc = 0; k.sk = SK_CHUNK; k.C2 = a.C2;
    const int B = a.batch.count;
    if (a.ntiles <= 0 || k.KT <= 0) return hipSuccess;
    if (!a.streamk) {   // one whole tile per workgroup
        if (a.nwg != a.ntiles || a.s || a.diag_pad_from >= 0) return hipErrorInvalidValue;
        if (a.tile_edge == 64 && k.KT == 8)      hipLaunchKernelGGL((gemm_nt_tile_k128_kernel<2, 2>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else if (a.tile_edge == 32 && k.KT == 8) hipLaunchKernelGGL(gemm_nt_rows32_k128_kernel, dim3(a.ntiles, 1, B), dim3(512), 0, st, k);
        else if (a.tile_edge == 3232 && k.KT == 8) hipLaunchKernelGGL((gemm_nt_tile_k128_kernel<1, 1>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else if (a.tile_edge == 64) hipLaunchKernelGGL((gemm_nt_tile_kernel<2, 2>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else if (a.tile_edge == 3232) hipLaunchKernelGGL((gemm_nt_tile_kernel<1, 1>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else if (a.tile_edge == 32) hipLaunchKernelGGL((gemm_nt_tile_kernel<1, 4>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        else                        hipLaunchKernelGGL((gemm_nt_tile_kernel<4, 4>), dim3(a.ntiles, 1, B), dim3(256), 0, st, k);
        return hipGetLastError();
    }
    // canonical chunked summation (see the head of this file).  Data-parallel tiles take beta at their first chunk and
    // add alpha * chunk afterwards; stream-K tiles get alpha and beta in the fix-up.
    if (!a.sk_claim) return hipErrorInvalidValue;
    if (a.ldp >= (1 << 22) || a.ldq >= (1 << 22) || a.ldc >= (1 << 22)) return hipErrorInvalidValue;   // 128-row panels are 32-bit buffers
    if (a.s && a.ldp != a.ldq) return hipErrorInvalidValue;    // the scaled form is A.D.A^T: both operands are A
    k.kc = gemm_streamk_chunk(k.KT);
    if (k.KT <= k.kc) k.kc = 0;              // short contraction: one running sum per tile (and SK_CHUNK >= KT: whole tiles)
    if (k.kc == 0 && k.KT > SK_CHUNK) return hipErrorInvalidValue;
    k.sk = k.kc == 0 ? SK_CHUNK : streamk_unit(k.KT);
    const int cpt = streamk_cpt(k.KT);
    const int nrem = a.ntiles - (a.ntiles / a.nwg) * a.nwg;
    if (a.C2 && (cpt == 1 || a.beta != 0.0)) return hipErrorInvalidValue;   // the second copy comes from a tile's last flush / the fix-up
    if (nrem > 0) {
        if (cpt > 1 && !a.ws) return hipErrorInvalidValue;
        unsigned int* claim = (unsigned int*)((char*)a.sk_claim + (size_t)a.batch.first * (size_t)a.batch.stride);
        hipError_t em = B == 1 ? hipMemsetAsync(claim, 0, sizeof(unsigned int), st)
                               : hipMemset2DAsync(claim, (size_t)a.batch.stride, 0, sizeof(unsigned int), (size_t)B, st);
        if (em != hipSuccess) return em;
    }
    // a batch of a multiple of 8 LPs: one LP per XCD at a time (see BatchK)
    const bool xm = B >= 8 && B % 8 == 0;
    k.bk.xcd_major = xm ? 1 : 0;
    const dim3 grid = xm ? dim3(8, a.nwg, B / 8) : dim3(a.nwg, 1, B);
    if (a.s) hipLaunchKernelGGL(gemm_nt_streamk_w8_kernel<true>, grid, dim3(512), 0, st, k);
    else     hipLaunchKernelGGL(gemm_nt_streamk_w8_kernel<false>, grid, dim3(512), 0, st, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (nrem > 0 && cpt > 1) {
        hipLaunchKernelGGL(gemm_nt_fixup_kernel, xm ? dim3(8, nrem * FIX_SPLIT, B / 8) : dim3(nrem * FIX_SPLIT, 1, B), dim3(256), 0,
                           st, k);
        e = hipGetLastError();
    }
    return e;
}

// The canonical chunking of the A.D.A^T contraction in the units kernel.  Up to KT = 256 k-tiles (n <= 4096): uniform chunks
// of gemm_streamk_chunk(KT) -- the chunking of the round-2 kernel, so that an LP gets the same bits from either (the sizes
// that run alone and as lockstep batches).  Above: chunks of 64 k-tiles (1024 columns), except that the LAST 64 are cut
// into pieces of 16: units are dispatched in chunk order, so the launch ends on quarter-size units -- with 4224 equal units
// on 512 slots the last quarter-full round of 0.25 ms units cost 0.19 ms of a 2.3 ms launch.
int adat_units_chunking(int K, int* kc_out, int* nbig_out, int* ks_out) {
    const int KT = K / BK, kc = gemm_streamk_chunk(KT);
    int nbig, ks, cpt;
    if (KT <= kc) { nbig = 1; ks = kc; cpt = 1; }
    else if (KT <= 256 || kc < 32) { ks = kc; cpt = (KT + kc - 1) / kc; nbig = cpt; }
    else {
        nbig = KT / kc - 1;                 // full chunks but the last one
        ks = kc / 4;
        cpt = nbig + (KT - nbig * kc + ks - 1) / ks;
    }
    if (kc_out) *kc_out = kc;
    if (nbig_out) *nbig_out = nbig;
    if (ks_out) *ks_out = ks;
    return cpt;
}
int adat_units_cpt(int K) { return adat_units_chunking(K, nullptr, nullptr, nullptr); }

hipError_t launch_adat_units(const AdatUnitsArgs& a, hipStream_t st) {
    if (a.ntiles <= 0 || a.nunits <= 0 || a.K <= 0) return hipSuccess;
    if (a.lda >= (1 << 22) || a.ldc >= (1 << 22)) return hipErrorInvalidValue;   // 128-row panels are 32-bit buffers
    if (!a.A || !a.s || !a.C || !a.tile_list || !a.unit_list || a.upc < 1) return hipErrorInvalidValue;
    UnitsK k{};
    k.A = a.A; k.lda = a.lda; k.s = a.s; k.C = a.C; k.ldc = a.ldc; k.C2 = a.C2;
    k.KT = a.K / BK;
    k.cpt = adat_units_chunking(a.K, &k.kc, &k.nbig, &k.ks);
    if (k.cpt == 1) k.kc = 0;
    if (a.upc > 1 && k.nbig != k.cpt) return hipErrorInvalidValue;     // several chunks per unit: uniform chunking only
    if (k.cpt > 1 && (!a.slabs || !a.tile_cnt)) return hipErrorInvalidValue;
    if (k.cpt > 256) return hipErrorInvalidValue;          // a tile's slabs are one 32-bit buffer
    k.ntiles = a.ntiles; k.tile_list = a.tile_list; k.unit_list = a.unit_list; k.nunits = a.nunits; k.upc = a.upc;
    k.diag_pad_from = a.diag_pad_from; k.slabs = a.slabs; k.tile_cnt = a.tile_cnt;
    k.grp_cnt = a.grp_cnt; k.grp_w = a.grp_w > 0 ? a.grp_w : 1; k.bk = batch_k(a.batch);
    const int B = a.batch.count;
    const bool xm = B >= 8 && B % 8 == 0;                  // one LP per XCD at a time (see BatchK)
    k.bk.xcd_major = xm ? 1 : 0;
    const dim3 grid = xm ? dim3(8, a.nunits, B / 8) : dim3(a.nunits, 1, B);
    if (a.grp_cnt) hipLaunchKernelGGL(gemm_nt_units_kernel<true>, grid, dim3(512), 0, st, k);
    else           hipLaunchKernelGGL(gemm_nt_units_kernel<false>, grid, dim3(512), 0, st, k);
    return hipGetLastError();
}

hipError_t launch_wait_count(const unsigned int* cnt, unsigned int target, const int* done, unsigned int* timeout, hipStream_t st) {
    hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(64), 0, st, cnt, target, done, timeout);
    return hipGetLastError();
}

hipError_t launch_gemm_grouped(const GemmTileDesc* descs_dev, int ntiles, hipStream_t st, const Batch& bt, int edge) {
    if (ntiles <= 0) return hipSuccess;
    if (edge == 32)      hipLaunchKernelGGL(gemm_nt_grouped32_kernel, dim3(ntiles, 1, bt.count), dim3(256), 0, st, descs_dev, batch_k(bt));
    else if (edge == 64) hipLaunchKernelGGL(gemm_nt_grouped64_kernel, dim3(ntiles, 1, bt.count), dim3(256), 0, st, descs_dev, batch_k(bt));
    else            hipLaunchKernelGGL(gemm_nt_grouped_kernel, dim3(ntiles, 1, bt.count), dim3(256), 0, st, descs_dev, batch_k(bt));
    return hipGetLastError();
}

}  // namespace lpipm
