// lpipm_internal.hpp -- shared declarations of liblpipm.so (host C++ + HIP kernels for gfx950).
// Product code: nothing here may include, link or call anything under oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <vector>
#include <utility>
#include "../../include/lpipm.h"

struct lpipm_ctx;

namespace lpipm {

// ---------------------------------------------------------------- geometry
constexpr int TILE = 128;  // output tile edge of the MFMA NT-GEMM == Cholesky block size NB
constexpr int BK   = 16;   // k-tile depth staged through LDS per barrier
constexpr int NB   = 128;  // Cholesky / TRSV block size (== TILE)

inline uint64_t round_up(uint64_t v, uint64_t q) { return (v + q - 1) / q * q; }

// ---------------------------------------------------------------- measurement knobs
// Every LPIPM_* environment variable the library reads is a MEASUREMENT knob (kernel variants, schedules, widths: several
// of them change the bits of a result).  They are read through this one function, which ignores them unless
// LPIPM_EXPERIMENTAL=1 is set as well: a stray variable in a production (or parity-test) environment cannot change what the
// library computes.  tests/conftest.py refuses to run with LPIPM_EXPERIMENTAL set; tests that exercise a knob set both.
const char* lp_knob(const char* name);

// ---------------------------------------------------------------- what another translation unit may know of a context
struct lpipm_ctx_device { int device; hipStream_t stream; };
lpipm_ctx_device lpipm_ctx_device_of(lpipm_ctx* ctx);      // solver.hip

// ---------------------------------------------------------------- error plumbing
void set_error_detail(const char* what, hipError_t e, const char* file, int line);
#define LP_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) {                                            \
            ::lpipm::set_error_detail(#expr, e__, __FILE__, __LINE__);      \
            return LPIPM_ERR_HIP;                                           \
        }                                                                   \
    } while (0)

// ---------------------------------------------------------------- lockstep batches
// `count` LPs of identical geometry live in identical arenas `stride` BYTES apart and advance in
// lockstep: every launch covers all of them through gridDim.z, a kernel of LP z shifts each of its
// pointers by z*stride.  done (nullable) is LP 0's "finished" word: an LP whose word is non-zero is
// skipped by every kernel, so its iterate stays frozen while the others go on.
// count == 1, stride == 0 is the ordinary single-LP launch.
struct Batch {
    int count = 1;
    long long stride = 0;
    const int* done = nullptr;
    int first = 0;        // a launch may cover only the LPs [first, first + count) of the resident batch (pointers stay LP 0's)
};
inline Batch batch_slice(const Batch& b, int first, int count) { return Batch{count, b.stride, b.done, b.first + first}; }
// The part a kernel needs.  xcd_major (A.D.A^T only, count a multiple of 8): grid = (8, workgroups per LP,
// count / 8) and LP = 8*blockIdx.z + blockIdx.x -- workgroups are dealt to the 8 XCDs round-robin along x,
// so all workgroups of one LP share one XCD's L2 and re-use each other's panels of A.
struct BatchK { long long stride; const int* done; int xcd_major; int first; };
inline BatchK batch_k(const Batch& b) { return BatchK{b.stride, b.done, 0, b.first}; }
#ifdef __HIPCC__
__device__ __forceinline__ long long batch_lp(const BatchK& b) {
    return (long long)b.first + (b.xcd_major ? (long long)blockIdx.z * 8 + blockIdx.x : (long long)blockIdx.z);
}
__device__ __forceinline__ bool batch_done(const BatchK& b) {
    return b.done && *(const int*)((const char*)b.done + batch_lp(b) * b.stride) != 0;
}
template <typename T>
__device__ __forceinline__ T* batch_ptr(T* p, const BatchK& b) {   // null stays null
    return p ? (T*)((char*)p + batch_lp(b) * b.stride) : p;
}
template <typename T>
__device__ __forceinline__ const T* batch_ptr(const T* p, const BatchK& b) {
    return p ? (const T*)((const char*)p + batch_lp(b) * b.stride) : p;
}
#endif

// Bump allocator over one arena (per-LP state).  A first pass with base == nullptr only measures.
struct Arena {
    char* base = nullptr;
    size_t off = 0;
    template <typename T>
    T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = (T*)((uintptr_t)base + off);
        off += (count ? count : 1) * sizeof(T);
        return p;
    }
};

// ---------------------------------------------------------------- NT GEMM (kernels_gemm.hip)
// C(tile ti,tj) = beta*C + alpha * sum_k P[ti*128+r][k] * s[k] * Q[tj*128+c][k]
// Row-major operands with K contiguous ("NT"): this one MFMA kernel serves
//   A.D.A^T            (P = Q = A, s = x/z, lower tiles, stream-K over n)   newton_equations.rs:55-57
//   trailing update    (P = Q = L21, alpha=-1, beta=1, lower tiles)         Cholesky, :129-131
//   TRSM as GEMM       (P = A21, Q = inv(L11), rectangular tiles)
struct GemmArgs {
    const double* P; int64_t ldp;
    const double* Q; int64_t ldq;
    const double* s;            // nullable: per-k scale applied to the Q panel while staging
    double*       C; int64_t ldc;
    int           K;            // multiple of BK
    double        alpha, beta;
    int           ntiles;
    int           tiles_lower;  // 1: tile index -> lower triangle (row-major), 0: rectangular
    int           ntj;          // rectangular: number of tile columns
    const int2*   tile_list;    // nullable: explicit (ti,tj) order (device pointer)
    int           diag_pad_from;// rows/cols >= this on the diagonal are written as 1.0 (-1: off)
    int           streamk;      // 1: the A.D.A^T kernel (data-parallel tiles + claimed stream-K chunks, canonical chunked
                                //    summation; alpha = 1, beta = 0); 0: whole-tile kernels, nwg == ntiles
    double*       ws;           // stream-K chunk slabs: gemm_streamk_slabs() tiles of TILE*TILE doubles
    int           nwg;          // workgroups launched per LP
    unsigned int* sk_claim;     // stream-K: the device word through which the chunks of the remainder tiles are claimed
                                //   (workgroups that finish their data-parallel tiles early take more of them); slabs are
                                //   indexed by chunk, so the sums do not depend on who computed what
    double*       C2;           // stream-K launches whose tiles are all split (ntiles < nwg): final values stored here too
    int           tile_edge;    // whole-tile launches: 0/128 -> 128x128 tiles, 64 -> 64x64, 32 -> 32 rows x 128 cols, 3232 -> 32x32
    Batch         batch;        // lockstep batch (every pointer above except tile_list is per LP)
};
// Launches the main kernel and, when tiles are split into stream-K chunks, the deterministic fix-up pass.
hipError_t launch_gemm_nt(const GemmArgs& a, hipStream_t st);
// stream-K geometry: canonical chunk (k-tiles) for a contraction of KT k-tiles, workgroup count for ntiles x KT
// work on num_cu CUs, and the number of TILE x TILE slabs ws must hold for a given workgroup count
int gemm_streamk_chunk(int KT);
size_t gemm_streamk_slabs(int ntiles, int KT, int nwg);
bool gemm_streamk_split(int KT);     // contraction long enough to be cut into chunks (only then can GemmArgs::C2 be used)
int gemm_streamk_nwg(int ntiles, int KT, int num_cu);
// ---- A.D.A^T as (tile, chunk) UNITS with an in-launch combine (kernels_gemm.hip, gemm_nt_units_kernel) --------------
// The contraction of every lower tile is cut into the canonical chunks (gemm_streamk_chunk); a workgroup computes `upc`
// consecutive chunks of ONE tile, writes each chunk sum to that chunk's slab (write-through stores: nothing to wait for),
// and adds the number of chunks it did to the tile's arrival counter.  The workgroup whose add completes the tile adds the
// tile's slabs IN CHUNK ORDER and stores the tile -- the same sums in the same order as a single running pass that flushes
// at every chunk boundary, whatever the decomposition (no separate fix-up launch; the bits of M do not depend on the
// workgroup count, on `upc`, or on who arrives last).  A non-persistent grid, one unit per workgroup, dispatched in list
// order: with a column-group-major list the groups of M complete one after the other WHILE the launch runs, and the
// workgroup that completes a group's last tile bumps that group's word -- what the factorisation's chain waits for
// (solver.hip, enqueue_factor_grouped).
struct AdatUnitsArgs {
    const double* A; int64_t lda;     // P = Q = A
    const double* s;                  // dinv (per-k scale)
    double* C; int64_t ldc;           // M
    double* C2;                       // nullable second copy
    int K;                            // columns (multiple of BK)
    int ntiles;
    const int2* tile_list;            // (ti, tj) per tile, device
    const int2* unit_list;            // (tile index, first chunk) per unit in dispatch order, device; tile < 0: padding (no-op)
    int nunits, upc;
    int diag_pad_from;
    double* slabs;                    // ntiles * cpt slabs of TILE*TILE doubles
    unsigned int* tile_cnt;           // ntiles arrival counters, ZEROED before the launch (by the caller, in stream order)
    unsigned int* grp_cnt;            // nullable: completed tiles per column group (tj / grp_w), zeroed likewise
    int grp_w;
    Batch batch;
};
int adat_units_cpt(int K);            // chunks per tile for a contraction of K columns
int adat_units_chunking(int K, int* kc, int* nbig, int* ks);   // ... and their boundaries (kernels_gemm.hip)
hipError_t launch_adat_units(const AdatUnitsArgs& a, hipStream_t st);
// One wave that returns when *cnt >= target (or when *done != 0, or after a bounded number of polls, which sets *timeout):
// the device-side wait of a stream for a group word of a running launch on another stream.
hipError_t launch_wait_count(const unsigned int* cnt, unsigned int target, const int* done, unsigned int* timeout, hipStream_t st);

// One output tile (128x128, or 64x64 with edge = 64) of a grouped launch: C = alpha * P[0:E, kb:ke) . Q[0:E, kb:ke)^T
// (k-range in units of BK), each tile with its own operands.
struct GemmTileDesc {
    const double* P; const double* Q; double* C;
    int ldp, ldq, ldc;
    int kt_begin, kt_end;
    double alpha;
};
// descs_dev holds LP 0's pointers; LP z of a batch shifts P, Q, C by z*stride.
hipError_t launch_gemm_grouped(const GemmTileDesc* descs_dev, int ntiles, hipStream_t st, const Batch& bt = Batch{},
                               int edge = 128);

// ---------------------------------------------------------------- factor plan (kernels_trsv.hip)
// The factor L is consumed through explicit inverses of its diagonal SUPER-blocks (up to 1024 wide):
// a triangular solve is then a handful of fully parallel mat-vec launches instead of mp/128
// serialized block steps.  The plan owns the inverse storage and the static launch descriptors.
constexpr int SUPER = 1024;  // default super-block width (multiple of NB)
struct SuperBlock {
    int row0, size;          // first row/column of the diagonal super-block, its width (multiple of NB)
    double* inv;             // size x size row-major: inv(L_ss)   (lower triangular, zeros above)
    double* invT;            // size x size row-major: inv(L_ss)^T (upper triangular, zeros below)
};
struct FactorPlan {
    int mp = 0;
    int merge_edge = 128;    // output tile edge of the merge GEMMs (32 / 64: latency-bound, 128: flop-bound)
    int super_w = SUPER;     // width of the diagonal super-blocks whose inverses are formed: wider = fewer, fully
                             // parallel solve steps, but the merge GEMMs cost flops (a batch that already fills
                             // the chip prefers 512)
    std::vector<SuperBlock> sbs;
    GemmTileDesc* descs_dev = nullptr;           // grouped-GEMM tiles of all merge stages (own allocation, shared by a batch)
    std::vector<std::pair<int, int>> stages;     // (first descriptor, count) per launch, in order
    double* tpart = nullptr;                     // gemv_t slabs of the backward sweep
    // 128-block k of the factorisation -> where its inverse / transposed inverse go, and their ld
    double* blk_inv(int k) const;
    double* blk_invT(int k) const;
    int blk_ld(int k) const;
};
// Takes the inverse storage for an mp x mp factor living at (L, ld) from the arena (which must be zeroed:
// the never-written halves of the triangular inverses are read as zeros) and, when `build`, uploads the
// merge descriptors.  build == false: sizing pass over a measuring arena, nothing is allocated.
hipError_t factor_plan_create(FactorPlan& plan, const double* L, int64_t ld, int mp, Arena& arena, bool build, hipStream_t st,
                              int super_w = SUPER, int merge_edge = 128);
void factor_plan_destroy(FactorPlan& plan);

// ---------------------------------------------------------------- Cholesky (kernels_potrf.hip)
// In-place blocked lower Cholesky of the mp x mp row-major matrix M (mp multiple of NB), followed by
// the inverses of the diagonal super-blocks (plan).  info (device int32): 0, or 1 + index of the
// first non-positive pivot.
// Look-ahead for the trailing updates of a single factorisation: `side` is a second stream (CU-masked, so that the chain
// always finds free CUs), ev_chain / ev_rest one event per outer panel.  nullptr: everything on `st`.
struct PotrfLookahead {
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev_chain, ev_rest;
    int min_nb = 32;         // 128-blocks from which the look-ahead is used (default: m >= 4096; LPIPM_LOOKAHEAD=1: 12)
};
hipError_t launch_potrf(double* M, int64_t ld, int mp, const FactorPlan& plan, int32_t* info, hipStream_t st,
                        const Batch& bt = Batch{}, const PotrfLookahead* la = nullptr, bool clear_info = true);

// The pieces of launch_potrf, for the factorisation that runs beside A.D.A^T (solver.hip, enqueue_factor_overlapped):
constexpr int POTRF_OUTER = 4;   // 128-blocks per outer panel
hipError_t potrf_clear_info(int32_t* info, hipStream_t st, const Batch& bt);
hipError_t potrf_panel_chain(double* M, int64_t ld, int mp, const FactorPlan& plan, int32_t* info, hipStream_t st,
                             const Batch& bt, int J0, int J1);
hipError_t potrf_trailing_update(double* M, int64_t ld, int mp, hipStream_t st, const Batch& bt, int J0, int J1);
hipError_t potrf_superblock_inverses(const FactorPlan& plan, hipStream_t st, const Batch& bt);

// ---------------------------------------------------------------- triangular solves (kernels_trsv.hip)
// R[r] <- L^-T L^-1 R[r], r < nrhs (1|2); R is nrhs x mp (row stride mp); Yscratch: nrhs x mp.
hipError_t launch_chol_solve(const double* L, int64_t ld, const FactorPlan& plan, int nrhs, double* R,
                             double* Yscratch, hipStream_t st, const Batch& bt = Batch{});

// ---------------------------------------------------------------- QR arms (kernels_qr.hip)
// EquationSolverType::{Inverse, LeastSquares}: Householder QR of the full mp x mp matrix whose LOWER
// triangle is in M (the upper one is filled in first); reflectors below the diagonal, R on and above,
// tau[mp].  info: 0, or 1 + index of a zero column / zero R diagonal.
hipError_t launch_qr_factor(double* M, int64_t ld, int mp, double* tau, int32_t* info, hipStream_t st);
// R[q] <- R^-1 Q^T R[q] in place, q < nrhs (row stride mp); mp <= 16384
hipError_t launch_qr_solve(const double* M, int64_t ld, int mp, const double* tau, int nrhs, double* R,
                           int32_t* info, hipStream_t st);

// ---------------------------------------------------------------- GEMV (kernels_gemv.hip)
// Y[r][i] = (add[r] ? add[r][i] : 0) + alpha * sum_k A[i][k] * W[r][k],   i < m, k < np
hipError_t launch_gemv_n(const double* A, int64_t lda, int m, int np, int nrhs, const double* W,
                         int64_t ldw, const double* add0, const double* add1, double* Y, int64_t ldy,
                         hipStream_t st, double alpha = 1.0, const Batch& bt = Batch{});
// Upart[s][r][k] = sum_{i in row split s} A[i][k] * V[r][i];  consumers sum the splits in order.
constexpr int GEMVT_ROWS = 128;
// np: columns processed (multiple of 2); slab: stride between slabs (0 = np)
hipError_t launch_gemv_t(const double* A, int64_t lda, int mp, int np, int nrhs, const double* V,
                         int64_t ldv, double* Upart, hipStream_t st, int64_t slab = 0, const Batch& bt = Batch{});
// One read of A for both products of the residual pair: AxPart[ch][i] = sum over column chunk ch of A[i][k] W[k]
// (gemv_dual_chunks(np) slabs of mp doubles: consumers add them in order) and the row-split slabs of A^T.V as gemv_t.
int gemv_dual_chunks(int np);
hipError_t launch_gemv_dual(const double* A, int64_t lda, int mp, int np, const double* W, const double* V, double* AxPart,
                            double* Upart, int64_t slab, hipStream_t st, const Batch& bt = Batch{});
// Rho[q] = R0[q] - M.V[q] (q < nrhs) for a symmetric mp x mp M whose LOWER triangle is stored (one read of it);
// slabs: symv_slab_doubles(mp) doubles of scratch.  The residual of the refinement step of the Cholesky solve.
size_t symv_slab_doubles(int mp);
hipError_t launch_symv_residual(const double* M, int64_t ld, int mp, int nrhs, const double* V, int64_t ldv, const double* R0,
                                int64_t ldr, double* Rho, int64_t ldo, double* slabs, hipStream_t st, const Batch& bt = Batch{});
// slack structure [I; 0] of the last ns columns (never stored), see kernels_gemv.hip
hipError_t launch_slack_n(int ns, int nx, int nrhs, const double* W, int64_t ldw, double* Y, int64_t ldy, hipStream_t st,
                          const Batch& bt = Batch{});
hipError_t launch_slack_t(int ns, int nx, int nrhs, int nsplit, const double* V, int64_t ldv, double* Upart, int64_t slab,
                          hipStream_t st, const Batch& bt = Batch{});
hipError_t launch_slack_diag(int ns, int nx, const double* d, double* M, int64_t ldm, hipStream_t st,
                             const Batch& bt = Batch{});
// U[r][k] = sum_s Upart[s][r][k]   (stand-alone reduce; the solver fuses this into its consumers)
hipError_t launch_gemv_t_reduce(const double* Upart, int nsplit, int nrhs, int np, double* U,
                                int64_t ldu, hipStream_t st);

// ---------------------------------------------------------------- probe (kernels_probe.hip)
hipError_t launch_mfma_probe(int iters, double* sink, int blocks, hipStream_t st);

}  // namespace lpipm
