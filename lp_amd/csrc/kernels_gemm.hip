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

namespace lpipm {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int LDS_STRIDE = BK + 2;  // doubles per LDS row (144 B, keeps 16-B alignment)

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
};

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

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmK p) {
    __shared__ __attribute__((aligned(16))) double lds[2][2][TILE][LDS_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int g = xcd_remap(blockIdx.x, gridDim.x);

    const int KT = p.KT;
    const long long total = (long long)p.ntiles * KT;
    long long it = wg_begin(g, total, p.nwg);
    const long long end = wg_begin(g + 1, total, p.nwg);
    bool first = true;

    const int srow = tid >> 3;        // staging: 32 rows per pass, 8 lanes per 128-B row segment
    const int scol = (tid & 7) * 2;

    while (it < end) {
        const int tile = (int)(it / KT);
        const int kb = (int)(it - (long long)tile * KT);
        const int ke = (int)((long long)(KT - kb) < (end - it) ? KT : kb + (end - it));
        int ti, tj;
        tile_coords(p, tile, ti, tj);
        const double* Pp = p.P + (long long)(ti * TILE + srow) * p.ldp + scol;
        const double* Qp = p.Q + (long long)(tj * TILE + srow) * p.ldq + scol;

        d4 acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = (d4){0.0, 0.0, 0.0, 0.0};

        d2 sa[4], sb[4];
        auto gload = [&](int kt) {
            const long long ko = (long long)kt * BK;
#pragma unroll
            for (int r = 0; r < 4; ++r) sa[r] = *(const d2*)(Pp + (long long)(32 * r) * p.ldp + ko);
#pragma unroll
            for (int r = 0; r < 4; ++r) sb[r] = *(const d2*)(Qp + (long long)(32 * r) * p.ldq + ko);
            if (p.s) {
                const d2 sv = *(const d2*)(p.s + ko + scol);
#pragma unroll
                for (int r = 0; r < 4; ++r) sb[r] = sb[r] * sv;
            }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *(d2*)&lds[buf][0][srow + 32 * r][scol] = sa[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) *(d2*)&lds[buf][1][srow + 32 * r][scol] = sb[r];
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
                d2 a[4], b[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    a[mi] = *(const d2*)&lds[cur][0][wr * 64 + mi * 16 + fr][round * 8 + fq * 2];
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
                    b[nj] = *(const d2*)&lds[cur][1][wc * 64 + nj * 16 + fr][round * 8 + fq * 2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int nj = 0; nj < 4; ++nj)
                            acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t],
                                                                                acc[mi][nj], 0, 0, 0);
            }
            if (more) lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }

        // ---- epilogue.  C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg.
        // Per-lane base pointer + wave-uniform row offsets keep the address math in SGPRs.
        const bool full = (kb == 0 && ke == KT);
        if (full) {
            double* cb = p.C + (long long)(ti * TILE + wr * 64 + fq) * p.ldc + (tj * TILE + wc * 64 + fr);
            const bool pad_diag = p.diag_pad_from >= 0 && ti == tj && wr == wc;
            const int row0 = ti * TILE + wr * 64 + fq;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* rp = cb + (long long)(mi * 16 + 4 * r) * p.ldc;
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj) {
                        double v = p.alpha * acc[mi][nj][r];
                        if (p.beta != 0.0) v += p.beta * rp[nj * 16];
                        if (pad_diag && mi == nj && fq + 4 * r == fr && row0 + mi * 16 + 4 * r >= p.diag_pad_from)
                            v = 1.0;
                        rp[nj * 16] = v;
                    }
                }
        } else {
            double* sb0 = p.ws + ((long long)(2 * g + (first ? 0 : 1))) * (TILE * TILE) +
                          (wr * 64 + fq) * TILE + wc * 64 + fr;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj)
                        sb0[(mi * 16 + 4 * r) * TILE + nj * 16] = acc[mi][nj][r];
        }
        it += ke - kb;
        first = false;
    }
}

// Adds the partial slabs of every tile whose k-range was split, in workgroup order.
__global__ __launch_bounds__(256) void gemm_nt_fixup_kernel(const GemmK p) {
    const int tile = blockIdx.x;
    const int KT = p.KT;
    const long long total = (long long)p.ntiles * KT;
    const long long it0 = (long long)tile * KT, it1 = it0 + KT;
    const int g_lo = wg_owner(it0, total, p.nwg), g_hi = wg_owner(it1 - 1, total, p.nwg);
    if (g_lo == g_hi) return;  // one workgroup computed the whole tile and wrote it directly
    int ti, tj;
    tile_coords(p, tile, ti, tj);
    for (int e = threadIdx.x * 2; e < TILE * TILE; e += 512) {
        d2 sum = (d2){0.0, 0.0};
        for (int g = g_lo; g <= g_hi; ++g) {
            const int slot = wg_begin(g, total, p.nwg) >= it0 ? 0 : 1;
            sum += *(const d2*)(p.ws + ((long long)(2 * g + slot)) * (TILE * TILE) + e);
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
    k.diag_pad_from = a.diag_pad_from; k.ws = a.ws; k.nwg = a.nwg;
    if (a.ntiles <= 0 || k.KT <= 0) return hipSuccess;
    hipLaunchKernelGGL(gemm_nt_kernel, dim3(a.nwg), dim3(256), 0, st, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // a split exists unless every workgroup boundary falls on a tile boundary
    const long long total = (long long)a.ntiles * k.KT;
    bool split = false;
    if (a.nwg != a.ntiles) {
        for (int g = 1; g < a.nwg && !split; ++g) split = ((g * total) / a.nwg) % k.KT != 0;
    }
    if (split) {
        hipLaunchKernelGGL(gemm_nt_fixup_kernel, dim3(a.ntiles), dim3(256), 0, st, k);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace lpipm
