// vec_kernels.hpp -- argument block, scalar-slot indices and launchers of kernels_vec.hip.
#pragma once
#include "lpipm_internal.hpp"

namespace lpipm {

constexpr int RED_STRIDE = 512;  // max workgroups of a vector kernel == partials per reduction slot
constexpr int RED_SLOTS  = 8;

// device scalar block S
enum {
    S_TAU = 0, S_KAPPA, S_MU, S_RG, S_GAMMA, S_ETA, S_RHAT_G, S_RHAT_TK, S_DTAU, S_DKAPPA,
    S_ALPHA_PRED, S_ALPHA, S_CP, S_BQ, S_RP0, S_RD0, S_RG0, S_RMU0,
    S_C0,   // the constant of the objective (written at upload; every LP of a batch has its own)
    S_COUNT
};
enum { ST_OPTIMAL = 0, ST_INFEASIBLE = 1, ST_UNBOUNDED = 2, ST_UNFINISHED = 3 };  // indicators.rs:85-90
enum { FLAG_NAN_PQ = 1 };
// Selective refinement (LPIPM_REFINE=1, a measurement mode; the default, LPIPM_REFINE unset, refines nothing; =2 refines every solve): the Cholesky solves of an
// iteration are refined when mu / mu_0 of the point the normal equations are formed at is at most this.  The decision is
// the LP's own (device word skip_refine, mirrored by the host from the status record): the same alone and in a batch.
constexpr double REFINE_BELOW_RHO_MU = 1e-2;
double refine_below();   // REFINE_BELOW_RHO_MU, or LPIPM_REFINE_BELOW from the environment (measurement knob)

// read back once per iteration (96 bytes)
struct StatusRec {
    double alpha, rho_p, rho_d, rho_A, rho_g, rho_mu, obj, tau, kappa;
    int32_t status, potrf_info, flags, pad_;
};

struct VecArgs {
    int n, m, np, mp, nblk, nsplit;
    long long n_total;   // columns of the WHOLE LP (== n unless the LP is split by columns over ranks)
    double* gs;          // n-split mode: globally reduced sums / mins the scalar kernels read instead of `red`
    // problem
    const double *b, *c;
    // iterate
    double *x, *y, *z;
    // work vectors (n-sized: np doubles; m-sized: mp doubles)
    double *dinv, *xs, *r1, *rD, *p, *u, *dx, *dz, *dxdz;   // n
    double *rP, *rP2, *q, *dy, *Ax;                          // m
    double *W;        // [2][np]  gemv_n inputs
    double *R;        // [2][mp]  gemv_n outputs / solve in-out
    const double* ATpart;  // [nsplit][nrhs][np] gemv_t slabs
    double *S, *red;
    StatusRec* status;
    StatusRec* status_pinned;   // nullable: the context's coherent pinned host array, one record per LP of the launch (dense, index
                                //   blockIdx.z).  The kernels that write `status` write the record here too -- payload, system
                                //   fence, then the sequence word pad_ -- so the host can wait for an iteration by watching pad_:
                                //   no D2H copy launch and no event record between the indicators and the next A.D.A^T
    int status_seq;             // what pad_ is set to by the launch this argument block goes to
    int32_t *potrf_info;
    int *flags;
    int *skip_refine; // set by k_scalar_indicators: 1 = the refinement step of this LP's Cholesky solves is skipped in the next
                      //   iteration (the LP has finished, or its normal equations are still well conditioned)
    int *done;        // set by k_scalar_indicators when the LP has reached a final status; cleared by k_blind_start
    const int *done_chk;  // what the kernels of the iteration test before doing anything (nullptr: no test)
    int ax_chunks;    // Ax holds this many slabs of mp doubles whose sum is A.x (1: a plain gemv_n result)
    double refine_below;  // threshold of skip_refine (refine_below())
    int bcount;       // lockstep batch: LPs per launch (gridDim.z); every pointer above is LP 0's,
    long long bstride;//   LP z's is bstride bytes * z further
    int bfirst;       // the launch covers the LPs [bfirst, bfirst + bcount) of the resident batch
};

// n-split mode (a.gs != nullptr): between a vector kernel and the scalar kernel that consumes its
// partial sums, the sums over the split dimension are folded into a.gs and handed to `xr` for the
// cross-rank reduction (op 0 = sum, 1 = min); xr == nullptr / a.gs == nullptr: single-GPU path.
struct XRank {
    int (*fn)(void* self, double* dev_ptr, int count, int op);
    void* self;
};
void vec_blind_start(const VecArgs& a, hipStream_t st);
// with_pred (only where vec_fused(a) holds and xr is null): the launch also does vec_pred_setup for the next iteration
int  vec_residuals(const VecArgs& a, int is_init, int ip_next, double tol, hipStream_t st, const XRank* xr = nullptr,
                   bool with_pred = false);
void vec_pred_setup(const VecArgs& a, hipStream_t st);
int  vec_pq_uv(const VecArgs& a, hipStream_t st, const XRank* xr = nullptr);
int  vec_uv_corr(const VecArgs& a, hipStream_t st, const XRank* xr = nullptr);
int  vec_delta(const VecArgs& a, int phase, int ip, double alpha0, hipStream_t st, const XRank* xr = nullptr);
void vec_corr_setup(const VecArgs& a, int ip, hipStream_t st);
void vec_step(const VecArgs& a, int ip, double alpha0, hipStream_t st);
// one launch for a run of the kernels above (kernels_vec.hip, "fused vector stage"): vec_fused(a) says whether they apply
bool vec_fused(const VecArgs& a);
void vec_fused_predictor(const VecArgs& a, int ip, hipStream_t st);                  // vec_pq_uv + vec_delta(0) + vec_corr_setup
void vec_fused_corrector(const VecArgs& a, int ip, double alpha0, hipStream_t st);   // vec_uv_corr + vec_delta(1) + vec_step
int  vec_final_x(const VecArgs& a, double* xout, hipStream_t st, const XRank* xr = nullptr);
// Y[q][i] += add_q[i] (i < m): the addend of a column-split A.w after its cross-rank sum
// packed has mp*(mp+128)/2 doubles; dir 0 = M -> packed, 1 = packed -> M (mp a multiple of 128, ld even)
void vec_pack_lower(double* M, long long ld, int mp, double* packed, int dir, hipStream_t st);
void vec_add_rows(int m, int nrhs, double* Y, long long ldy, const double* add0, const double* add1, hipStream_t st);
// tiles[t] = (ti, tj): packed[t][128][128] <-> the 128 x 128 tile of M at (ti, tj); dir 0 = M -> packed, 1 = packed -> M
void vec_pack_tiles(double* M, long long ld, const int2* tiles, int ntiles, double* packed, int dir, hipStream_t st);
// refined Cholesky solve (solver.hip): dst[q][0:mp] = / += src[q][0:mp]; M0 <- lower block-triangle of M
void vec_rows_copy(int mp, int nrhs, double* dst, const double* src, hipStream_t st, const Batch& bt = Batch{});
void vec_rows_add(int mp, int nrhs, double* dst, const double* src, hipStream_t st, const Batch& bt = Batch{});
void vec_copy_lower(const double* M, double* M0, long long ld, int mp, hipStream_t st, const Batch& bt = Batch{});


}  // namespace lpipm
