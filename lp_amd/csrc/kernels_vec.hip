// kernels_vec.hip -- the O(n) vector kernels and the one-wave scalar kernels of the IPM iteration.
//
// Everything the reference does between its GEMVs and solves (rhat.rs, delta.rs, the step-size
// ratio test and step of feasible_point.rs, residual.rs, indicators.rs) runs here on device so an
// iteration needs no host round trip: scalars (tau, kappa, mu, gamma, eta, d_tau, alpha, ...) live
// in a device block `S`, reductions are two-stage (per-workgroup partials in `red`, summed in
// fixed order by the next scalar kernel) and therefore bitwise reproducible.
// Each kernel cites the reference lines whose arithmetic (and operation order) it reproduces.
#include <cstdlib>
#include "vec_kernels.hpp"

namespace lpipm {

// ---------------------------------------------------------------- reduction helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
// workgroup (256 threads) reduction of K values; thread 0 writes red[slot0 + k][blockIdx.x]
template <int K, bool IS_MIN>
__device__ __forceinline__ void block_reduce_store(double (&v)[K], double* red, int slot0) {
    __shared__ double sm[4][K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double w = IS_MIN ? wave_min(v[k]) : wave_sum(v[k]);
        if (lane == 0) sm[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double r = IS_MIN ? fmin(fmin(sm[0][k], sm[1][k]), fmin(sm[2][k], sm[3][k]))
                                    : (sm[0][k] + sm[1][k]) + (sm[2][k] + sm[3][k]);
            red[(slot0 + k) * RED_STRIDE + blockIdx.x] = r;
        }
    }
}
// one wave folds the nblk partials of a slot in a fixed order; every lane gets the result
__device__ __forceinline__ double fold_sum(const double* red, int slot, int nblk) {
    double s = 0.0;
    for (int b = (int)(threadIdx.x & 63); b < nblk; b += 64) s += red[slot * RED_STRIDE + b];
    return wave_sum(s);
}
__device__ __forceinline__ double fold_min(const double* red, int slot, int nblk, double init) {
    double s = init;
    for (int b = (int)(threadIdx.x & 63); b < nblk; b += 64) s = fmin(s, red[slot * RED_STRIDE + b]);
    return wave_min(s);
}

// LP blockIdx.z of a lockstep batch.  `check_done`: kernels of the iteration skip an LP that has finished.
__device__ __forceinline__ bool vbatch(VecArgs& a, bool check_done) {
    const BatchK bk{a.bstride, check_done ? a.done_chk : nullptr, 0, a.bfirst};
    if (batch_done(bk)) return false;
    if (blockIdx.z == 0 && a.bfirst == 0) return true;
    a.b = batch_ptr(a.b, bk); a.c = batch_ptr(a.c, bk);
    a.x = batch_ptr(a.x, bk); a.y = batch_ptr(a.y, bk); a.z = batch_ptr(a.z, bk);
    a.dinv = batch_ptr(a.dinv, bk); a.xs = batch_ptr(a.xs, bk); a.r1 = batch_ptr(a.r1, bk); a.rD = batch_ptr(a.rD, bk);
    a.p = batch_ptr(a.p, bk); a.u = batch_ptr(a.u, bk); a.dx = batch_ptr(a.dx, bk); a.dz = batch_ptr(a.dz, bk);
    a.dxdz = batch_ptr(a.dxdz, bk);
    a.rP = batch_ptr(a.rP, bk); a.rP2 = batch_ptr(a.rP2, bk); a.q = batch_ptr(a.q, bk); a.dy = batch_ptr(a.dy, bk);
    a.Ax = batch_ptr(a.Ax, bk); a.W = batch_ptr(a.W, bk); a.R = batch_ptr(a.R, bk); a.ATpart = batch_ptr(a.ATpart, bk);
    a.S = batch_ptr(a.S, bk); a.red = batch_ptr(a.red, bk); a.status = batch_ptr(a.status, bk);
    a.potrf_info = batch_ptr(a.potrf_info, bk); a.flags = batch_ptr(a.flags, bk); a.done = batch_ptr(a.done, bk);
    a.skip_refine = batch_ptr(a.skip_refine, bk);
    a.done_chk = batch_ptr(a.done_chk, bk);
    return true;
}

// n-split mode: gs[first .. first+count) <- fold of the reduction slots (sum or min), optionally the NaN
// flag as a number in gs[flag_slot]; the host then reduces gs across ranks.
__global__ void k_fold(VecArgs a, int first, int count, int is_min, int flag_slot) {
    for (int s = first; s < first + count; ++s) {
        const double v = is_min ? fold_min(a.red, s, a.nblk, 1.0) : fold_sum(a.red, s, a.nblk);
        if (threadIdx.x == 0) a.gs[s] = v;
    }
    if (flag_slot >= 0 && threadIdx.x == 0) a.gs[flag_slot] = (*a.flags & FLAG_NAN_PQ) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(256) void k_add_rows(int m, int nrhs, double* Y, long long ldy, const double* add0,
                                                  const double* add1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    if (add0) Y[i] += add0[i];
    if (nrhs > 1 && add1) Y[ldy + i] += add1[i];
}

// Lower block-triangle of M (128-row block i keeps columns [0, 128(i+1))) <-> one contiguous buffer, so
// that the cross-rank sum of the partial normal equations moves m(m+128)/2 doubles instead of m^2.
// One workgroup per row; dir 0 = pack, 1 = unpack.
__global__ __launch_bounds__(256) void k_pack_lower(double* __restrict__ M, long long ld, double* __restrict__ P, int dir) {
    const int row = blockIdx.x, bi = row >> 7;
    const long long w = (long long)(bi + 1) * 128;
    double2* p = reinterpret_cast<double2*>(P + 8192ll * bi * (bi + 1) + (long long)(row & 127) * w);
    double2* q = reinterpret_cast<double2*>(M + (long long)row * ld);
    if (dir == 0) for (int j = threadIdx.x; j < w / 2; j += 256) p[j] = q[j];
    else          for (int j = threadIdx.x; j < w / 2; j += 256) q[j] = p[j];
}

// ---------------------------------------------------------------- FeasiblePoint::blind_start
// feasible_point.rs:24-31: x = 1, y = 0, z = 1, tau = kappa = 1
__global__ __launch_bounds__(256) void k_blind_start(VecArgs a) {
    if (!vbatch(a, false)) return;
    const int stride = gridDim.x * 256;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < a.n; j += stride) { a.x[j] = 1.0; a.z[j] = 1.0; }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.m; i += stride) a.y[i] = 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.S[S_TAU] = 1.0;
        a.S[S_KAPPA] = 1.0;
        *a.flags = 0;
        *a.done = 0;
        *a.potrf_info = 0;
    }
}

// ---------------------------------------------------------------- residuals at the current point
// residual.rs:22-31 and feasible_point.rs:122-123 (the same vectors):
//   r_P = b*tau - A.x            (Ax from gemv_n)
//   r_D = c*tau - A^T.y - z      (A^T.y = sum of the gemv_t row-split slabs)
// partial sums: |r_P|^2, b.y, |r_D|^2, c.x, x.z, c.(x/tau)   (indicators.rs:41-44)
// A vector kernel's work is written once, for the thread `vt` of the (virtual) 256-thread block `vb` of `nvb`: the plain
// kernels pass (blockIdx.x, threadIdx.x, gridDim.x); the fused single-workgroup kernels further down walk the same
// virtual blocks four at a time and rebuild the same reduction tree, so their sums have the same bits.
struct VThread { int vb, vt, nvb; };
__device__ __forceinline__ VThread plain_thread() { return VThread{(int)blockIdx.x, (int)threadIdx.x, (int)gridDim.x}; }

__device__ __forceinline__ void body_residuals(const VecArgs& a, const VThread t, double (&acc)[6]) {
    const int stride = t.nvb * 256;
    const double tau = a.S[S_TAU];
    for (int i = t.vb * 256 + t.vt; i < a.m; i += stride) {
        double ax = a.Ax[i];
        for (int ch = 1; ch < a.ax_chunks; ++ch) ax += a.Ax[(long long)ch * a.mp + i];
        const double r = a.b[i] * tau - ax;
        a.rP[i] = r;
        acc[0] += r * r;
        acc[1] += a.b[i] * a.y[i];
    }
    for (int j = t.vb * 256 + t.vt; j < a.n; j += stride) {
        double aty = 0.0;
        for (int s = 0; s < a.nsplit; ++s) aty += a.ATpart[(long long)s * a.np + j];
        const double xj = a.x[j], zj = a.z[j], cj = a.c[j];
        const double r = cj * tau - aty - zj;
        a.rD[j] = r;
        acc[2] += r * r;
        acc[3] += cj * xj;
        acc[4] += xj * zj;
        acc[5] += cj * (xj / tau);
    }
}
__global__ __launch_bounds__(256) void k_residuals(VecArgs a) {
    if (!vbatch(a, true)) return;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    body_residuals(a, plain_thread(), acc);
    block_reduce_store<6, false>(acc, a.red, 0);
}

// Residuals::calculate + Indicators::from_point_and_problem + Indicators::status
// (residual.rs:33-43, indicators.rs:37-55, :57-83), then the scalars the NEXT get_delta starts
// from (feasible_point.rs:119-125, rhat.rs:31,33).
struct NextDelta { double gamma, mu, eta; int finished; };
__device__ __forceinline__ NextDelta scalar_indicators(const VecArgs& a, int is_init, int ip_next, double tol, double rp2, double by,
                                                       double rd2, double cx, double xz, double cxt) {    // ONE thread
    double* S = a.S;
    const double tau = S[S_TAU], kappa = S[S_KAPPA];
    const double rho_p = sqrt(rp2);                                   // residual.rs:34
    const double rho_d = sqrt(rd2);                                   // residual.rs:35
    const double rho_g = fabs(kappa + cx - by);                       // residual.rs:27-29,36
    const double rho_mu = (xz + tau * kappa) / (double)(a.n_total + 1);  // residual.rs:30-32,37
    if (is_init) {                                                    // feasible_point.rs:32
        S[S_RP0] = rho_p; S[S_RD0] = rho_d; S[S_RG0] = rho_g; S[S_RMU0] = rho_mu;
    }
    StatusRec* st = a.status;
    const double obj = cxt + S[S_C0];                                 // indicators.rs:41
    const double bty = by;                                            // indicators.rs:42
    const double rho_A = fabs(cx - bty) / (tau + fabs(by));           // indicators.rs:43-44
    const double ip_ = rho_p / fmax(S[S_RP0], 1.0);                   // indicators.rs:47
    const double id_ = rho_d / fmax(S[S_RD0], 1.0);                   // indicators.rs:48
    const double ig_ = rho_g / fmax(S[S_RG0], 1.0);                   // indicators.rs:50
    const double imu = rho_mu / S[S_RMU0];                            // indicators.rs:51
    st->alpha = is_init ? 1.0 : S[S_ALPHA];
    st->rho_p = ip_; st->rho_d = id_; st->rho_A = rho_A; st->rho_g = ig_; st->rho_mu = imu; st->obj = obj;
    st->tau = tau; st->kappa = kappa;
    int status = ST_UNFINISHED;
    if (!is_init) {                                                   // indicators.rs:66-83
        const bool tau_too_small = tau < tol * fmax(kappa, 1.0);
        const bool inf1 = (ip_ < tol && id_ < tol && ig_ < tol) && tau_too_small;
        const bool inf2 = imu < tol && tau_too_small;
        if (inf1 || inf2) status = bty > tol ? ST_INFEASIBLE : ST_UNBOUNDED;
        else if (ip_ < tol && id_ < tol && rho_A < tol) status = ST_OPTIMAL;
    }
    st->status = status;
    st->potrf_info = *a.potrf_info;
    st->flags = *a.flags;
    // what ends the loop of solve_normal_form (mod.rs:215, :231-233): from here on the LP's kernels are skipped
    const bool finished = !is_init && (status != ST_UNFINISHED || *a.potrf_info != 0 || (*a.flags & FLAG_NAN_PQ));
    if (finished) *a.done = 1;
    *a.potrf_info = 0;      // read and recorded: the next factorisation starts from a clean word (launch_potrf need not clear it)
    *a.skip_refine = (finished || !(imu <= a.refine_below)) ? 1 : 0;
    // next get_delta (feasible_point.rs:119-125)
    const double gamma = ip_next ? 1.0 : 0.0;
    const double eta = ip_next ? 1.0 : 1.0 - gamma;
    const double rG = cx - by + kappa;                                // :124
    const double mu = (xz + tau * kappa) / (double)(a.n_total + 1);   // :125
    S[S_RG] = rG; S[S_MU] = mu; S[S_GAMMA] = gamma; S[S_ETA] = eta;
    S[S_RHAT_G] = rG * eta;                                           // rhat.rs:31
    S[S_RHAT_TK] = gamma * mu - tau * kappa;                          // rhat.rs:33
    st->pad_ = a.status_seq;
    if (a.status_pinned) {       // the record straight into the host's array: payload, system fence, then the sequence word
        StatusRec* h = a.status_pinned + blockIdx.z;
        h->alpha = st->alpha; h->rho_p = st->rho_p; h->rho_d = st->rho_d; h->rho_A = st->rho_A; h->rho_g = st->rho_g;
        h->rho_mu = st->rho_mu; h->obj = st->obj; h->tau = st->tau; h->kappa = st->kappa;
        h->status = st->status; h->potrf_info = st->potrf_info; h->flags = st->flags;
        __threadfence_system();
        __hip_atomic_store(&h->pad_, (int32_t)a.status_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return NextDelta{gamma, mu, eta, finished ? 1 : 0};
}
__global__ void k_scalar_indicators(VecArgs a, int is_init, int ip_next, double tol) {
    if (!vbatch(a, true)) return;
    const int nblk = a.nblk;
    // |r_P|^2 and b.y run over m (replicated on every rank); the other four over the (possibly split) n
    const double rp2 = fold_sum(a.red, 0, nblk), by = fold_sum(a.red, 1, nblk);
    const double rd2 = a.gs ? a.gs[2] : fold_sum(a.red, 2, nblk), cx = a.gs ? a.gs[3] : fold_sum(a.red, 3, nblk);
    const double xz = a.gs ? a.gs[4] : fold_sum(a.red, 4, nblk), cxt = a.gs ? a.gs[5] : fold_sum(a.red, 5, nblk);
    if (threadIdx.x != 0) return;
    (void)scalar_indicators(a, is_init, ip_next, tol, rp2, by, rd2, cx, xz, cxt);
}

// ---------------------------------------------------------------- predictor set-up
// newton_equations.rs:54 (Dinv = x/z); rhat.rs:29-32 (predictor r_hat); the r1 argument of the
// second sym_solve (newton_equations.rs:188: rhat.d - rhat.xs/x) and the Dinv*r1 prologues of both
// sym_solve calls (:220).
__device__ __forceinline__ void body_pred_setup(const VecArgs& a, const VThread t, double gm, double eta) {
    const int stride = t.nvb * 256;
    for (int j = t.vb * 256 + t.vt; j < a.n; j += stride) {
        const double xj = a.x[j], zj = a.z[j];
        const double dinv = xj / zj;
        const double xs = (xj * -1.0) * zj + gm;
        const double r1 = a.rD[j] * eta - xs / xj;
        a.dinv[j] = dinv;
        a.xs[j] = xs;
        a.r1[j] = r1;
        a.W[j] = dinv * a.c[j];
        a.W[a.np + j] = dinv * r1;
    }
}
__global__ __launch_bounds__(256) void k_pred_setup(VecArgs a) {
    if (!vbatch(a, true)) return;
    body_pred_setup(a, plain_thread(), a.S[S_GAMMA] * a.S[S_MU], a.S[S_ETA]);
}

// sym_solve epilogue u = Dinv*(A^T.v - r1) for both solves of the predictor
// (newton_equations.rs:223), the four dots of delta.rs:29-32 and the NaN check of :190-194.
__device__ __forceinline__ int body_pq_uv(const VecArgs& a, const VThread t, double (&acc)[4]) {
    const int stride = t.nvb * 256;
    int nan = 0;
    for (int j = t.vb * 256 + t.vt; j < a.n; j += stride) {
        double atq = 0.0, atv = 0.0;
        for (int s = 0; s < a.nsplit; ++s) {
            atq += a.ATpart[((long long)s * 2 + 0) * a.np + j];
            atv += a.ATpart[((long long)s * 2 + 1) * a.np + j];
        }
        const double d = a.dinv[j], cj = a.c[j];
        const double p = d * (atq - cj);
        const double u = d * (atv - a.r1[j]);
        a.p[j] = p;
        a.u[j] = u;
        acc[0] += cj * p;
        acc[1] += cj * u;
        nan |= (p != p);
    }
    for (int i = t.vb * 256 + t.vt; i < a.m; i += stride) {
        const double q = a.R[i], v = a.R[a.mp + i], bi = a.b[i];
        a.q[i] = q;
        acc[2] += bi * q;
        acc[3] += bi * v;
        nan |= (q != q);
    }
    return nan;
}
__global__ __launch_bounds__(256) void k_pq_uv(VecArgs a) {
    if (!vbatch(a, true)) return;
    double acc[4] = {0, 0, 0, 0};
    if (body_pq_uv(a, plain_thread(), acc)) atomicOr(a.flags, FLAG_NAN_PQ);
    block_reduce_store<4, false>(acc, a.red, 0);
}

// corrector: only (u, v) change; (p, q) are identical to the predictor's (the reference recomputes
// them, feasible_point.rs:149 -> newton_equations.rs:187, with the same inputs and factor).
__device__ __forceinline__ void body_uv_corr(const VecArgs& a, const VThread t, double (&acc)[2]) {
    const int stride = t.nvb * 256;
    for (int j = t.vb * 256 + t.vt; j < a.n; j += stride) {
        double atv = 0.0;
        for (int s = 0; s < a.nsplit; ++s) atv += a.ATpart[(long long)s * a.np + j];
        const double u = a.dinv[j] * (atv - a.r1[j]);
        a.u[j] = u;
        acc[0] += a.c[j] * u;
    }
    for (int i = t.vb * 256 + t.vt; i < a.m; i += stride) acc[1] += a.b[i] * a.R[i];
}
__global__ __launch_bounds__(256) void k_uv_corr(VecArgs a) {
    if (!vbatch(a, true)) return;
    double acc[2] = {0, 0};
    body_uv_corr(a, plain_thread(), acc);
    block_reduce_store<2, false>(acc, a.red, 0);
}

// ---- the scalar steps as device functions: each is run either by its own one-wave kernel (column-split mode: a cross-rank
// reduction sits between the vector kernel and the scalar step) or FOLDED into the vector kernel that consumes its result
// (single GPU): every workgroup of the consumer folds the partial sums itself (fixed order: identical bits in every
// workgroup), workgroup 0 leaves the results in S for the kernels behind it.  One dependent launch less per fold (~4.5 us)
// for ~2 us of dependent loads in front of the consumer.  Rules that keep the folds race-free: a folded kernel never
// reads an S entry that its own workgroup 0 writes, and never writes a reduction slot that workgroups of the SAME launch
// still read (k_delta's minima therefore go to slots 4, 5 when folded: the dots it reads sit in 0 .. 3).
struct DtauOut { double d_tau, d_kappa, cp, bq; };
__device__ __forceinline__ DtauOut dtau_from(const VecArgs& a, double cp, double cu, double bq, double bv) {   // delta.rs:29-32, :38
    const double* S = a.S;
    const double tau = S[S_TAU], kappa = S[S_KAPPA];
    DtauOut o;
    o.d_tau = (S[S_RHAT_G] + 1.0 / tau * S[S_RHAT_TK] - (-cu + bv)) / (1.0 / tau * kappa + (-cp + bq));
    o.d_kappa = 1.0 / tau * (S[S_RHAT_TK] - kappa * o.d_tau);
    o.cp = cp; o.bq = bq;
    return o;
}
__device__ __forceinline__ DtauOut scalar_dtau(const VecArgs& a, int phase) {   // (no gs: single GPU)
    const int nblk = a.nblk;
    double cp, cu, bq, bv;
    if (phase == 0) { cp = fold_sum(a.red, 0, nblk); cu = fold_sum(a.red, 1, nblk); bq = fold_sum(a.red, 2, nblk); bv = fold_sum(a.red, 3, nblk); }
    else            { cu = fold_sum(a.red, 0, nblk); bv = fold_sum(a.red, 1, nblk); cp = a.S[S_CP]; bq = a.S[S_BQ]; }
    return dtau_from(a, cp, cu, bq, bv);
}
// get_step_size tail (feasible_point.rs:63-71) from the minima over x and z
__device__ __forceinline__ double amin_from(const VecArgs& a, double ax, double az, double d_tau, double d_kappa) {
    const double* S = a.S;
    const double tau = S[S_TAU], kappa = S[S_KAPPA];
    const double at = d_tau < 0.0 ? fmin(1.0, tau / -d_tau) : 1.0;
    const double ak = d_kappa < 0.0 ? fmin(1.0, kappa / -d_kappa) : 1.0;
    return fmin(fmin(fmin(fmin(1.0, ax), at), az), ak);
}
__device__ __forceinline__ double scalar_amin(const VecArgs& a, int mslot) {     // from the folded minima in slots mslot, mslot + 1
    const double ax = fold_min(a.red, mslot, a.nblk, 1.0), az = fold_min(a.red, mslot + 1, a.nblk, 1.0);
    return amin_from(a, ax, az, a.S[S_DTAU], a.S[S_DKAPPA]);
}
struct CorrScal { double alpha, gamma, eta, tk; };
__device__ __forceinline__ CorrScal corr_from(const VecArgs& a, double amin, int ip, double d_tau, double d_kappa) {   // feasible_point.rs:134-136,156-165; rhat.rs:51-74
    const double* S = a.S;
    const double tau = S[S_TAU], kappa = S[S_KAPPA], mu = S[S_MU];
    CorrScal c;
    c.alpha = amin * 1.0;
    if (ip) c.gamma = 10.0;
    else c.gamma = (1.0 - c.alpha) * (1.0 - c.alpha) * fmin(0.1, 1.0 - c.alpha);
    c.eta = ip ? 1.0 : 1.0 - c.gamma;
    if (ip) { const double alpha_2 = c.alpha * c.alpha; c.tk = (1.0 - c.alpha) * c.gamma * mu - tau * kappa - alpha_2 * d_tau * d_kappa; }
    else c.tk = c.gamma * mu - tau * kappa - d_tau * d_kappa;
    return c;
}
__device__ __forceinline__ CorrScal scalar_corr(const VecArgs& a, double amin, int ip) {
    return corr_from(a, amin, ip, a.S[S_DTAU], a.S[S_DKAPPA]);
}

// delta.rs:29-32 (d_tau) and :38 (d_kappa).  phase 0: predictor (all four dots fresh);
// phase 1: corrector (c.p, b.q reused from the predictor).
__global__ void k_scalar_dtau(VecArgs a, int phase) {
    if (!vbatch(a, true)) return;
    const int nblk = a.nblk;
    double cp, cu, bq, bv;
    if (phase == 0) {   // c.p, c.u run over the (possibly split) n; b.q, b.v over m (replicated)
        cp = a.gs ? a.gs[0] : fold_sum(a.red, 0, nblk); cu = a.gs ? a.gs[1] : fold_sum(a.red, 1, nblk);
        bq = fold_sum(a.red, 2, nblk); bv = fold_sum(a.red, 3, nblk);
        if (a.gs && a.gs[2] > 0.0 && threadIdx.x == 0) atomicOr(a.flags, FLAG_NAN_PQ);   // some rank saw NaN in p
    } else {
        cu = a.gs ? a.gs[0] : fold_sum(a.red, 0, nblk); bv = fold_sum(a.red, 1, nblk);
        cp = a.S[S_CP]; bq = a.S[S_BQ];
    }
    if (threadIdx.x != 0) return;
    double* S = a.S;
    const double tau = S[S_TAU], kappa = S[S_KAPPA];
    const double d_tau = (S[S_RHAT_G] + 1.0 / tau * S[S_RHAT_TK] - (-cu + bv)) /
                         (1.0 / tau * kappa + (-cp + bq));
    const double d_kappa = 1.0 / tau * (S[S_RHAT_TK] - kappa * d_tau);
    S[S_CP] = cp; S[S_BQ] = bq;
    S[S_DTAU] = d_tau; S[S_DKAPPA] = d_kappa;
}

// delta.rs:33-37 + the folds of get_step_size (feasible_point.rs:54-62).
// phase 0 keeps only d_x*d_z (all the corrector needs, rhat.rs:55,64); phase 1 keeps d_x, d_y, d_z.
__device__ __forceinline__ void body_delta(const VecArgs& a, const VThread t, int phase, double d_tau, double (&mn)[2]) {
    const int stride = t.nvb * 256;
    for (int j = t.vb * 256 + t.vt; j < a.n; j += stride) {
        const double xj = a.x[j], zj = a.z[j];
        const double dx = a.u[j] + a.p[j] * d_tau;
        const double dz = (a.xs[j] - zj * dx) / xj;
        if (dx < 0.0) mn[0] = fmin(mn[0], xj / -dx);
        if (dz < 0.0) mn[1] = fmin(mn[1], zj / -dz);
        if (phase == 0) a.dxdz[j] = dx * dz;
        else { a.dx[j] = dx; a.dz[j] = dz; }
    }
    if (phase == 1)
        for (int i = t.vb * 256 + t.vt; i < a.m; i += stride)
            a.dy[i] = a.R[i] + a.q[i] * d_tau;
}
template <bool FOLD>
__global__ __launch_bounds__(256) void k_delta(VecArgs a, int phase) {
    if (!vbatch(a, true)) return;
    double d_tau;
    if (FOLD) {     // k_scalar_dtau folded in
        const DtauOut o = scalar_dtau(a, phase);
        d_tau = o.d_tau;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.S[S_DTAU] = o.d_tau; a.S[S_DKAPPA] = o.d_kappa;
            if (phase == 0) { a.S[S_CP] = o.cp; a.S[S_BQ] = o.bq; }      // (phase 1 READS them: not rewritten there)
        }
    } else d_tau = a.S[S_DTAU];
    double mn[2] = {1.0, 1.0};
    body_delta(a, plain_thread(), phase, d_tau, mn);
    block_reduce_store<2, true>(mn, a.red, FOLD ? 4 : 0);
}

// get_step_size tail (feasible_point.rs:63-71).  phase 0: alpha of the predictor (alpha0 = 1),
// update_gamma (:156-165), eta (:136) and the scalar parts of Rhat::corrector (rhat.rs:51-74).
// phase 1: the step length of the iteration (interior_point/mod.rs:216-221).
__global__ void k_scalar_alpha(VecArgs a, int phase, int ip, double alpha0) {
    if (!vbatch(a, true)) return;
    const int nblk = a.nblk;
    const double ax = a.gs ? a.gs[0] : fold_min(a.red, 0, nblk, 1.0), az = a.gs ? a.gs[1] : fold_min(a.red, 1, nblk, 1.0);
    if (threadIdx.x != 0) return;
    double* S = a.S;
    const double tau = S[S_TAU], kappa = S[S_KAPPA], d_tau = S[S_DTAU], d_kappa = S[S_DKAPPA];
    const double at = d_tau < 0.0 ? fmin(1.0, tau / -d_tau) : 1.0;
    const double ak = d_kappa < 0.0 ? fmin(1.0, kappa / -d_kappa) : 1.0;
    const double amin = fmin(fmin(fmin(fmin(1.0, ax), at), az), ak);
    if (phase == 0) {
        const double alpha = amin * 1.0;                              // feasible_point.rs:134
        const double mu = S[S_MU];
        double gamma;
        if (ip) gamma = 10.0;                                         // :158-160
        else gamma = (1.0 - alpha) * (1.0 - alpha) * fmin(0.1, 1.0 - alpha);  // :163-164
        const double eta = ip ? 1.0 : 1.0 - gamma;                    // :136
        double tk;
        if (ip) {                                                     // rhat.rs:52,57-59
            const double alpha_2 = alpha * alpha;
            tk = (1.0 - alpha) * gamma * mu - tau * kappa - alpha_2 * d_tau * d_kappa;
        } else {                                                      // rhat.rs:65
            tk = gamma * mu - tau * kappa - d_tau * d_kappa;
        }
        S[S_ALPHA_PRED] = alpha; S[S_GAMMA] = gamma; S[S_ETA] = eta;
        S[S_RHAT_G] = S[S_RG] * eta;                                  // rhat.rs:71
        S[S_RHAT_TK] = tk;
    } else {
        S[S_ALPHA] = ip ? 1.0 : amin * alpha0;                        // mod.rs:216-221
    }
}

// Rhat::corrector vector parts (rhat.rs:51-56 / :62-64, :69-70) and the r1 / Dinv*r1 of the
// corrector's sym_solve (newton_equations.rs:188, :220).
__device__ __forceinline__ void body_corr_setup(const VecArgs& a, const VThread t, int ip, double gamma, double eta, double alpha,
                                                double mu) {
    const int stride = t.nvb * 256;
    const double alpha_2 = alpha * alpha;
    const double ipterm = (1.0 - alpha) * gamma * mu;
    const double gm = gamma * mu;
    for (int j = t.vb * 256 + t.vt; j < a.n; j += stride) {
        const double xj = a.x[j], zj = a.z[j], pr = a.dxdz[j];
        double xs;
        if (ip) xs = (xj * -1.0) * zj - pr * alpha_2 + ipterm;
        else    xs = (xj * -1.0) * zj + gm - pr;
        const double r1 = a.rD[j] * eta - xs / xj;
        a.xs[j] = xs;
        a.r1[j] = r1;
        a.W[j] = a.dinv[j] * r1;
    }
    for (int i = t.vb * 256 + t.vt; i < a.m; i += stride) a.rP2[i] = a.rP[i] * eta;
}
template <bool FOLD>
__global__ __launch_bounds__(256) void k_corr_setup(VecArgs a, int ip) {
    if (!vbatch(a, true)) return;
    double gamma, eta, alpha;
    const double mu = a.S[S_MU];
    if (FOLD) {     // k_scalar_alpha(phase 0) folded in
        const CorrScal c = scalar_corr(a, scalar_amin(a, 4), ip);
        gamma = c.gamma; eta = c.eta; alpha = c.alpha;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.S[S_ALPHA_PRED] = c.alpha; a.S[S_GAMMA] = c.gamma; a.S[S_ETA] = c.eta;
            a.S[S_RHAT_G] = a.S[S_RG] * c.eta;                           // rhat.rs:71
            a.S[S_RHAT_TK] = c.tk;
        }
    } else { gamma = a.S[S_GAMMA]; eta = a.S[S_ETA]; alpha = a.S[S_ALPHA_PRED]; }
    body_corr_setup(a, plain_thread(), ip, gamma, eta, alpha, mu);
}

// FeasiblePoint::do_step (feasible_point.rs:76-106)
__device__ __forceinline__ void body_step(const VecArgs& a, const VThread t, int ip, double alpha) {
    const int stride = t.nvb * 256;
    for (int j = t.vb * 256 + t.vt; j < a.n; j += stride) {
        double xn = a.x[j] + a.dx[j] * alpha;
        double zn = a.z[j] + a.dz[j] * alpha;
        if (ip) { xn = fmax(xn, 1.0); zn = fmax(zn, 1.0); }
        a.x[j] = xn;
        a.z[j] = zn;
    }
    for (int i = t.vb * 256 + t.vt; i < a.m; i += stride) a.y[i] = a.y[i] + a.dy[i] * alpha;
}
template <bool FOLD>
__global__ __launch_bounds__(256) void k_step(VecArgs a, int ip, double alpha0) {
    if (!vbatch(a, true)) return;
    double alpha;
    if (FOLD) {     // k_scalar_alpha(phase 1) folded in: the step length of the iteration (mod.rs:216-221)
        alpha = ip ? 1.0 : scalar_amin(a, 4) * alpha0;
        if (blockIdx.x == 0 && threadIdx.x == 0) a.S[S_ALPHA] = alpha;   // (tau, kappa move in k_step_scalars, behind this launch)
    } else alpha = a.S[S_ALPHA];
    body_step(a, plain_thread(), ip, alpha);
}
// tau / kappa part of do_step: separate one-thread launch so that no kernel both reads and writes S
__global__ void k_step_scalars(VecArgs a, int ip) {
    if (!vbatch(a, true)) return;
    if (threadIdx.x != 0) return;
    double* S = a.S;
    const double alpha = S[S_ALPHA];
    double tau = S[S_TAU] + S[S_DTAU] * alpha;
    double kappa = S[S_KAPPA] + S[S_DKAPPA] * alpha;
    if (ip) { tau = fmax(tau, 1.0); kappa = fmax(kappa, 1.0); }
    S[S_TAU] = tau;
    S[S_KAPPA] = kappa;
}

// ---------------------------------------------------------------- fused vector stage (one workgroup per LP)
// For SMALL LPs (FUSED_USE_NBLK virtual blocks: n, m <= 1024) the runs of vector kernels between two passes over A are ONE launch of
// one 1024-thread workgroup per LP: the workgroup walks the virtual 256-thread blocks four at a time (a wave is one of a
// block's four), and what separated the kernels -- a grid-wide reduction -- is a workgroup reduction with the SAME tree:
// butterfly inside each (virtual) wave, (w0 + w1) + (w2 + w3) per block, block sums into lanes 0 .. nblk-1 of a wave
// (0.0 + r, as fold_sum starts from zero) and the butterfly again.  Every thread handles the same elements in every phase,
// so what a phase stores for the next one is read back by the thread that wrote it.  The iterates are bit-identical to
// the kernel-by-kernel path (tests/test_gpu_solve.py, LPIPM_VEC_FUSED=0 as the other side).  What it saves is dependent
// launches: 3 -> 1 behind the predictor's passes, 4 -> 1 behind the corrector's, 2 -> 1 behind the residual pass.
constexpr int FUSED_THREADS = 1024;
constexpr int FUSED_MAX_NBLK = 32;    // what the kernels can do
constexpr int FUSED_USE_NBLK = 4;     // where they are used: n, m <= 1024.  One workgroup also pays every global round trip on
                                      // its own: vector stage per iteration 0.070 -> 0.057 ms at 512x1024, but 0.102 -> 0.113
                                      // at 32 x (1024x2048) and 0.094 -> 0.383 at 4096x8192 (32 row-split slabs of A^T.v per
                                      // column through one CU)
template <int K> struct FusedSm { double w[K][FUSED_MAX_NBLK][4]; };
__device__ __forceinline__ VThread fused_thread(const VecArgs& a, int round) {
    return VThread{4 * round + (int)(threadIdx.x >> 8), (int)(threadIdx.x & 255), a.nblk};
}
// the partial of this thread's (virtual) wave for each of K values -> sm.w
template <int K, bool IS_MIN>
__device__ __forceinline__ void fused_wave_part(const double (&v)[K], FusedSm<K>& sm, int vb) {
    const int lane = threadIdx.x & 63, vw = (threadIdx.x >> 6) & 3;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double w = IS_MIN ? wave_min(v[k]) : wave_sum(v[k]);
        if (lane == 0) sm.w[k][vb][vw] = w;
    }
}
// all parts in: every thread gets the K totals (same tree as block_reduce_store + fold_sum / fold_min: every wave forms the
// block sums of the blocks its lanes stand for and folds them itself -- ONE barrier; a FusedSm is used for one reduction only)
template <int K, bool IS_MIN>
__device__ __forceinline__ void fused_total(const FusedSm<K>& sm, int nblk, double (&out)[K]) {
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = IS_MIN ? 1.0 : 0.0;
        if (lane < nblk) {
            const double* w = sm.w[k][lane];
            const double r = IS_MIN ? fmin(fmin(w[0], w[1]), fmin(w[2], w[3])) : (w[0] + w[1]) + (w[2] + w[3]);
            s = IS_MIN ? fmin(s, r) : s + r;
        }
        out[k] = IS_MIN ? wave_min(s) : wave_sum(s);
    }
}

// k_pq_uv -> d_tau (delta.rs:29-32,38) -> k_delta(0) -> alpha, gamma, eta (feasible_point.rs:134-136) -> k_corr_setup
__global__ __launch_bounds__(FUSED_THREADS) void k_fused_predictor(VecArgs a, int ip) {
    if (!vbatch(a, true)) return;
    __shared__ FusedSm<4> sm4;
    __shared__ FusedSm<2> sm2;
    const int rounds = (a.nblk + 3) / 4;
    int nan = 0;
    for (int r = 0; r < rounds; ++r) {
        const VThread t = fused_thread(a, r);
        if (t.vb >= a.nblk) continue;                      // (whole waves)
        double acc[4] = {0, 0, 0, 0};
        nan |= body_pq_uv(a, t, acc);
        fused_wave_part<4, false>(acc, sm4, t.vb);
    }
    if (nan) atomicOr(a.flags, FLAG_NAN_PQ);
    double dots[4];
    fused_total<4, false>(sm4, a.nblk, dots);
    const DtauOut o = dtau_from(a, dots[0], dots[1], dots[2], dots[3]);
    for (int r = 0; r < rounds; ++r) {
        const VThread t = fused_thread(a, r);
        if (t.vb >= a.nblk) continue;
        double mn[2] = {1.0, 1.0};
        body_delta(a, t, 0, o.d_tau, mn);
        fused_wave_part<2, true>(mn, sm2, t.vb);
    }
    double mins[2];
    fused_total<2, true>(sm2, a.nblk, mins);
    const CorrScal c = corr_from(a, amin_from(a, mins[0], mins[1], o.d_tau, o.d_kappa), ip, o.d_tau, o.d_kappa);
    const double mu = a.S[S_MU], rg = a.S[S_RG];
    __syncthreads();                                       // every read of S above precedes the writes below
    if (threadIdx.x == 0) {
        a.S[S_DTAU] = o.d_tau; a.S[S_DKAPPA] = o.d_kappa; a.S[S_CP] = o.cp; a.S[S_BQ] = o.bq;
        a.S[S_ALPHA_PRED] = c.alpha; a.S[S_GAMMA] = c.gamma; a.S[S_ETA] = c.eta;
        a.S[S_RHAT_G] = rg * c.eta;                        // rhat.rs:71
        a.S[S_RHAT_TK] = c.tk;
    }
    for (int r = 0; r < rounds; ++r) {
        const VThread t = fused_thread(a, r);
        if (t.vb >= a.nblk) continue;
        body_corr_setup(a, t, ip, c.gamma, c.eta, c.alpha, mu);
    }
}

// k_uv_corr -> d_tau -> k_delta(1) -> the step length (mod.rs:216-221) -> do_step (feasible_point.rs:76-106) incl. tau, kappa
__global__ __launch_bounds__(FUSED_THREADS) void k_fused_corrector(VecArgs a, int ip, double alpha0) {
    if (!vbatch(a, true)) return;
    __shared__ FusedSm<2> sm2, sm2b;
    const int rounds = (a.nblk + 3) / 4;
    for (int r = 0; r < rounds; ++r) {
        const VThread t = fused_thread(a, r);
        if (t.vb >= a.nblk) continue;
        double acc[2] = {0, 0};
        body_uv_corr(a, t, acc);
        fused_wave_part<2, false>(acc, sm2, t.vb);
    }
    double dots[2];
    fused_total<2, false>(sm2, a.nblk, dots);
    const DtauOut o = dtau_from(a, a.S[S_CP], dots[0], a.S[S_BQ], dots[1]);
    for (int r = 0; r < rounds; ++r) {
        const VThread t = fused_thread(a, r);
        if (t.vb >= a.nblk) continue;
        double mn[2] = {1.0, 1.0};
        body_delta(a, t, 1, o.d_tau, mn);
        fused_wave_part<2, true>(mn, sm2b, t.vb);
    }
    double mins[2];
    fused_total<2, true>(sm2b, a.nblk, mins);
    const double alpha = ip ? 1.0 : amin_from(a, mins[0], mins[1], o.d_tau, o.d_kappa) * alpha0;
    double tau = a.S[S_TAU] + o.d_tau * alpha;             // k_step_scalars
    double kappa = a.S[S_KAPPA] + o.d_kappa * alpha;
    if (ip) { tau = fmax(tau, 1.0); kappa = fmax(kappa, 1.0); }
    __syncthreads();                                       // every read of S above precedes the writes below
    if (threadIdx.x == 0) {
        a.S[S_DTAU] = o.d_tau; a.S[S_DKAPPA] = o.d_kappa; a.S[S_ALPHA] = alpha;
        a.S[S_TAU] = tau; a.S[S_KAPPA] = kappa;
    }
    for (int r = 0; r < rounds; ++r) {
        const VThread t = fused_thread(a, r);
        if (t.vb >= a.nblk) continue;
        body_step(a, t, ip, alpha);
    }
}

// k_residuals -> k_scalar_indicators [-> k_pred_setup of the next iteration, unless the LP has just finished]
__global__ __launch_bounds__(FUSED_THREADS) void k_fused_residuals(VecArgs a, int is_init, int ip_next, double tol, int with_pred) {
    if (!vbatch(a, true)) return;
    __shared__ FusedSm<6> sm6;
    __shared__ NextDelta nd;
    const int rounds = (a.nblk + 3) / 4;
    for (int r = 0; r < rounds; ++r) {
        const VThread t = fused_thread(a, r);
        if (t.vb >= a.nblk) continue;
        double acc[6] = {0, 0, 0, 0, 0, 0};
        body_residuals(a, t, acc);
        fused_wave_part<6, false>(acc, sm6, t.vb);
    }
    double tot[6];
    fused_total<6, false>(sm6, a.nblk, tot);               // (behind its barrier: every thread has read S[S_TAU])
    if (threadIdx.x == 0) nd = scalar_indicators(a, is_init, ip_next, tol, tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]);
    if (!with_pred) return;
    __syncthreads();
    if (nd.finished) return;                               // k_pred_setup would have found the done word set
    const double gm = nd.gamma * nd.mu, eta = nd.eta;
    for (int r = 0; r < rounds; ++r) {
        const VThread t = fused_thread(a, r);
        if (t.vb >= a.nblk) continue;
        body_pred_setup(a, t, gm, eta);
    }
}

// x / tau (interior_point/mod.rs:231,238) and the partials of fun = c.(x/tau) (linear_program.rs:61-63)
__global__ __launch_bounds__(256) void k_final_x(VecArgs a, double* xout) {
    if (!vbatch(a, false)) return;
    xout = batch_ptr(xout, BatchK{a.bstride, nullptr, 0, a.bfirst});
    const int stride = gridDim.x * 256;
    const double tau = a.S[S_TAU];
    double acc[1] = {0};
    for (int j = blockIdx.x * 256 + threadIdx.x; j < a.n; j += stride) {
        const double v = a.x[j] / tau;
        xout[j] = v;
        acc[0] += a.c[j] * v;
    }
    block_reduce_store<1, false>(acc, a.red, 0);
}
__global__ void k_scalar_fun(VecArgs a) {
    if (!vbatch(a, false)) return;
    const double s = a.gs ? a.gs[0] : fold_sum(a.red, 0, a.nblk);
    if (threadIdx.x == 0) {
        a.status->obj = s + a.S[S_C0];
        if (a.status_pinned) { a.status_pinned[blockIdx.z].obj = s + a.S[S_C0]; __threadfence_system(); }   // (the host synchronises the stream behind this)
    }
}

// ---------------------------------------------------------------- launchers
static inline dim3 vgrid(const VecArgs& a) { return dim3(a.nblk, 1, a.bcount); }
static inline dim3 sgrid(const VecArgs& a) { return dim3(1, 1, a.bcount); }

void vec_blind_start(const VecArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_blind_start, vgrid(a), dim3(256), 0, st, a); }
static int cross(const VecArgs& a, const XRank* xr, int first, int count, int is_min, int flag_slot, int red_first,
                 int red_count, hipStream_t st) {
    if (!a.gs || !xr) return 0;
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(64), 0, st, a, first, count, is_min, flag_slot);
    return xr->fn(xr->self, a.gs + red_first, red_count, is_min);
}
double refine_below() {
    static const double v = lp_knob("LPIPM_REFINE_BELOW") ? atof(lp_knob("LPIPM_REFINE_BELOW")) : REFINE_BELOW_RHO_MU;
    return v;
}
// the fused single-workgroup kernels: single GPU, n and m within FUSED_MAX_NBLK virtual blocks (LPIPM_VEC_FUSED=0: never)
bool vec_fused(const VecArgs& a) {
    if (a.gs != nullptr || a.nblk > FUSED_USE_NBLK) return false;
    const char* e = lp_knob("LPIPM_VEC_FUSED");          // (read per call: the tests switch it inside one process)
    return !(e && e[0] == '0');
}
int vec_residuals(const VecArgs& a, int is_init, int ip_next, double tol, hipStream_t st, const XRank* xr, bool with_pred) {
    if (vec_fused(a) && !xr) {
        hipLaunchKernelGGL(k_fused_residuals, sgrid(a), dim3(FUSED_THREADS), 0, st, a, is_init, ip_next, tol, with_pred ? 1 : 0);
        return 0;
    }
    hipLaunchKernelGGL(k_residuals, vgrid(a), dim3(256), 0, st, a);
    if (int rc = cross(a, xr, 2, 4, 0, -1, 2, 4, st)) return rc;     // |r_D|^2, c.x, x.z, c.(x/tau)
    hipLaunchKernelGGL(k_scalar_indicators, sgrid(a), dim3(64), 0, st, a, is_init, ip_next, tol);
    return 0;
}
void vec_pred_setup(const VecArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_pred_setup, vgrid(a), dim3(256), 0, st, a); }
// Single GPU (a.gs == nullptr): the one-wave scalar kernels k_scalar_dtau / k_scalar_alpha are folded into their consumers
// (k_delta, k_corr_setup, k_step): four dependent launches less per iteration.  Column-split mode keeps them: the cross-rank
// reductions sit between a vector kernel and its scalar step.
static inline bool folded(const VecArgs& a) { return a.gs == nullptr; }
int vec_pq_uv(const VecArgs& a, hipStream_t st, const XRank* xr) {
    hipLaunchKernelGGL(k_pq_uv, vgrid(a), dim3(256), 0, st, a);
    if (folded(a)) return 0;                                          // d_tau: folded into k_delta(0)
    if (int rc = cross(a, xr, 0, 2, 0, 2, 0, 3, st)) return rc;      // c.p, c.u and the NaN-in-p flag (gs[2])
    hipLaunchKernelGGL(k_scalar_dtau, sgrid(a), dim3(64), 0, st, a, 0);
    return 0;
}
int vec_uv_corr(const VecArgs& a, hipStream_t st, const XRank* xr) {
    hipLaunchKernelGGL(k_uv_corr, vgrid(a), dim3(256), 0, st, a);
    if (folded(a)) return 0;                                          // d_tau: folded into k_delta(1)
    if (int rc = cross(a, xr, 0, 1, 0, -1, 0, 1, st)) return rc;     // c.u
    hipLaunchKernelGGL(k_scalar_dtau, sgrid(a), dim3(64), 0, st, a, 1);
    return 0;
}
int vec_delta(const VecArgs& a, int phase, int ip, double alpha0, hipStream_t st, const XRank* xr) {
    if (folded(a)) {                                                  // alpha: folded into k_corr_setup (phase 0) / k_step (phase 1)
        hipLaunchKernelGGL(k_delta<true>, vgrid(a), dim3(256), 0, st, a, phase);
        return 0;
    }
    hipLaunchKernelGGL(k_delta<false>, vgrid(a), dim3(256), 0, st, a, phase);
    if (int rc = cross(a, xr, 0, 2, 1, -1, 0, 2, st)) return rc;     // ratio-test minima over x and z
    hipLaunchKernelGGL(k_scalar_alpha, sgrid(a), dim3(64), 0, st, a, phase, ip, alpha0);
    return 0;
}
void vec_corr_setup(const VecArgs& a, int ip, hipStream_t st) {
    if (folded(a)) hipLaunchKernelGGL(k_corr_setup<true>, vgrid(a), dim3(256), 0, st, a, ip);
    else           hipLaunchKernelGGL(k_corr_setup<false>, vgrid(a), dim3(256), 0, st, a, ip);
}
void vec_fused_predictor(const VecArgs& a, int ip, hipStream_t st) {
    hipLaunchKernelGGL(k_fused_predictor, sgrid(a), dim3(FUSED_THREADS), 0, st, a, ip);
}
void vec_fused_corrector(const VecArgs& a, int ip, double alpha0, hipStream_t st) {
    hipLaunchKernelGGL(k_fused_corrector, sgrid(a), dim3(FUSED_THREADS), 0, st, a, ip, alpha0);
}
void vec_step(const VecArgs& a, int ip, double alpha0, hipStream_t st) {
    if (folded(a)) hipLaunchKernelGGL(k_step<true>, vgrid(a), dim3(256), 0, st, a, ip, alpha0);
    else           hipLaunchKernelGGL(k_step<false>, vgrid(a), dim3(256), 0, st, a, ip, alpha0);
    hipLaunchKernelGGL(k_step_scalars, sgrid(a), dim3(64), 0, st, a, ip);
}
int vec_final_x(const VecArgs& a, double* xout, hipStream_t st, const XRank* xr) {
    hipLaunchKernelGGL(k_final_x, vgrid(a), dim3(256), 0, st, a, xout);
    if (int rc = cross(a, xr, 0, 1, 0, -1, 0, 1, st)) return rc;     // c.(x/tau)
    hipLaunchKernelGGL(k_scalar_fun, sgrid(a), dim3(64), 0, st, a);
    return 0;
}
// One column group of M (its tiles, in the grouped tile list's order) <-> a contiguous block, for the group-by-group
// cross-rank sum of the column-split mode.  One workgroup per tile.
__global__ __launch_bounds__(256) void k_pack_tiles(double* __restrict__ M, long long ld, const int2* __restrict__ tiles,
                                                    double* __restrict__ P, int dir) {
    const int2 t = tiles[blockIdx.x];
    double* m0 = M + (long long)t.x * 128 * ld + (long long)t.y * 128;
    double2* p = reinterpret_cast<double2*>(P + (long long)blockIdx.x * 16384);
    for (int e = threadIdx.x; e < 8192; e += 256) {       // 128 rows x 64 pairs
        double2* q = reinterpret_cast<double2*>(m0 + (long long)(e >> 6) * ld) + (e & 63);
        if (dir == 0) p[e] = *q; else *q = p[e];
    }
}
void vec_pack_tiles(double* M, long long ld, const int2* tiles, int ntiles, double* packed, int dir, hipStream_t st) {
    if (ntiles > 0) hipLaunchKernelGGL(k_pack_tiles, dim3(ntiles), dim3(256), 0, st, M, ld, tiles, packed, dir);
}
void vec_pack_lower(double* M, long long ld, int mp, double* packed, int dir, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_lower, dim3(mp), dim3(256), 0, st, M, ld, packed, dir);
}
// batch-aware helpers of the refined Cholesky solve: dst[q][i] (op)= src[q][i], q < nrhs, i < mp; and the copy of the
// lower block-triangle of the normal-equations matrix that the refinement's residual is taken against
__global__ __launch_bounds__(256) void k_rows_op(int mp, int nrhs, double* __restrict__ dst, const double* __restrict__ src, int add,
                                                 BatchK bk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= mp || batch_done(bk)) return;
    dst = batch_ptr(dst, bk); src = batch_ptr(src, bk);
    for (int q = 0; q < nrhs; ++q) {
        const long long e = (long long)q * mp + i;
        dst[e] = add ? dst[e] + src[e] : src[e];
    }
}
void vec_rows_copy(int mp, int nrhs, double* dst, const double* src, hipStream_t st, const Batch& bt) {
    hipLaunchKernelGGL(k_rows_op, dim3((mp + 255) / 256, 1, bt.count), dim3(256), 0, st, mp, nrhs, dst, src, 0, batch_k(bt));
}
void vec_rows_add(int mp, int nrhs, double* dst, const double* src, hipStream_t st, const Batch& bt) {
    hipLaunchKernelGGL(k_rows_op, dim3((mp + 255) / 256, 1, bt.count), dim3(256), 0, st, mp, nrhs, dst, src, 1, batch_k(bt));
}
// grid (row, 128-column block): block-row bi keeps column blocks 0..bi
__global__ __launch_bounds__(64) void k_copy_lower(const double* __restrict__ M, double* __restrict__ M0, long long ld, BatchK bk) {
    if (batch_done(bk)) return;
    const int row = blockIdx.x, cb = blockIdx.y;
    if (cb > (row >> 7)) return;
    M = batch_ptr(M, bk); M0 = batch_ptr(M0, bk);
    const long long e = (long long)row * ld + cb * 128 + 2 * threadIdx.x;
    *(double2*)(M0 + e) = *(const double2*)(M + e);
}
void vec_copy_lower(const double* M, double* M0, long long ld, int mp, hipStream_t st, const Batch& bt) {
    hipLaunchKernelGGL(k_copy_lower, dim3(mp, mp / 128, bt.count), dim3(64), 0, st, M, M0, ld, batch_k(bt));
}

void vec_add_rows(int m, int nrhs, double* Y, long long ldy, const double* add0, const double* add1, hipStream_t st) {
    hipLaunchKernelGGL(k_add_rows, dim3((m + 255) / 256), dim3(256), 0, st, m, nrhs, Y, ldy, add0, add1);
}

}  // namespace lpipm
