// kernels_trsv.hip -- v = L^-T (L^-1 r) with the blocked factor of kernels_potrf.hip
// (replaces `factor.solvec_into(b2)`, newton_equations.rs:151-169; called from sym_solve :221).
//
// A substitution sweep over mp/128 blocks is mp/128 dependent steps, each a tiny launch: latency,
// not bytes.  Instead the factor is consumed through the explicit inverses of its diagonal
// SUPER-blocks (SUPER = 1024 wide; built once per factorisation from the 128-block inverses by
// log2(8) = 3 doubling levels of Inv21 = -Inv22.L21.Inv11 on the MFMA grouped GEMM):
//   forward  (k = 0..nsb-1):  y_k = Inv_k . r_k ;      r_below -= L[below, k] . y_k
//   backward (k = nsb-1..0):  y_k -= L[below, k]^T . v_below ;  v_k = Inv_k^T . y_k
// Every step is a row-split mat-vec over >= 128 workgroups (no serial inner dependency), so a solve
// is ~4*nsb launches that stream the lower triangle of L plus the inverses once each (HBM-bound).
// 1 or 2 right-hand sides per sweep (the predictor's two sym_solve calls share one).
// Reductions are fixed-order: results are bitwise reproducible.
#include "lpipm_internal.hpp"

namespace lpipm {

// ------------------------------------------------------------------------------------------------
// plan: storage + static descriptors of the merge GEMMs
double* FactorPlan::blk_inv(int k) const {
    const int r = k * NB;
    for (const SuperBlock& s : sbs)
        if (r >= s.row0 && r < s.row0 + s.size) return s.inv + (size_t)(r - s.row0) * s.size + (r - s.row0);
    return nullptr;
}
double* FactorPlan::blk_invT(int k) const {
    const int r = k * NB;
    for (const SuperBlock& s : sbs)
        if (r >= s.row0 && r < s.row0 + s.size) return s.invT + (size_t)(r - s.row0) * s.size + (r - s.row0);
    return nullptr;
}
int FactorPlan::blk_ld(int k) const {
    const int r = k * NB;
    for (const SuperBlock& s : sbs)
        if (r >= s.row0 && r < s.row0 + s.size) return s.size;
    return 0;
}

namespace {
struct Merge { int sb, lo, mid, hi, level; size_t toff; };   // block indices local to the super-block

// inverse of blocks [lo, hi) from the inverses of [lo, mid) and [mid, hi); returns its level
int plan_merges(int sb, int lo, int hi, std::vector<Merge>& out, size_t& tcursor) {
    if (hi - lo <= 1) return 0;
    int p = 1;
    while (p * 2 < hi - lo) p *= 2;   // left part: largest power of two below the width
    const int mid = lo + p;
    const int l1 = plan_merges(sb, lo, mid, out, tcursor);
    const int l2 = plan_merges(sb, mid, hi, out, tcursor);
    const int lvl = (l1 > l2 ? l1 : l2) + 1;
    out.push_back(Merge{sb, lo, mid, hi, lvl, tcursor});
    tcursor += (size_t)(mid - lo) * NB * (size_t)(hi - mid) * NB;
    return lvl;
}
}  // namespace

hipError_t factor_plan_create(FactorPlan& plan, const double* L, int64_t ld, int mp, Arena& arena, bool build,
                              hipStream_t st, int super_w, int merge_edge) {
    factor_plan_destroy(plan);
    plan.mp = mp;
    plan.super_w = super_w;
    plan.merge_edge = (merge_edge == 64 || merge_edge == 32) ? merge_edge : 128;
    const int E = plan.merge_edge, SUB = NB / E, EK = E / BK;   // sub-tiles per 128-block edge, k-tiles per sub-tile edge
    hipError_t e;
    for (int r0 = 0; r0 < mp; r0 += super_w) {
        SuperBlock s{};
        s.row0 = r0;
        s.size = mp - r0 < super_w ? mp - r0 : super_w;
        s.inv = arena.take<double>((size_t)s.size * s.size);
        s.invT = arena.take<double>((size_t)s.size * s.size);
        plan.sbs.push_back(s);
    }
    std::vector<Merge> merges;
    size_t tcursor = 0;
    int maxlvl = 0;
    for (size_t i = 0; i < plan.sbs.size(); ++i) {
        const int lvl = plan_merges((int)i, 0, plan.sbs[i].size / NB, merges, tcursor);
        if (lvl > maxlvl) maxlvl = lvl;
    }
    double* tws = arena.take<double>(tcursor);
    // slabs of the backward sweep's transposed panel products: (rows below / 128) x 2 rhs x SUPER
    plan.tpart = arena.take<double>((size_t)(mp / GEMVT_ROWS + 1) * 2 * super_w);
    if (!build) return hipSuccess;

    std::vector<GemmTileDesc> descs;
    plan.stages.clear();
    const int KPB = NB / BK;   // k-tiles per 128-block
    for (int lvl = 1; lvl <= maxlvl; ++lvl) {
        // stage A: T^T (s1 x s2) = Inv11^T . L21^T ; Inv11^T is upper triangular: k >= row tile
        const int a0 = (int)descs.size();
        for (const Merge& m : merges) {
            if (m.level != lvl) continue;
            const SuperBlock& s = plan.sbs[m.sb];
            const int n1 = m.mid - m.lo, n2 = m.hi - m.mid;
            for (int tj = 0; tj < n1 * SUB; ++tj)           // sub-tile indices: rows of T^T, columns of T^T
                for (int ti = 0; ti < n2 * SUB; ++ti) {
                    GemmTileDesc d{};
                    d.P = s.invT + (size_t)(m.lo * NB + tj * E) * s.size + m.lo * NB;
                    d.ldp = s.size;
                    d.Q = L + (size_t)(s.row0 + m.mid * NB + ti * E) * ld + s.row0 + m.lo * NB;
                    d.ldq = (int)ld;
                    d.C = tws + m.toff + (size_t)(tj * E) * (n2 * NB) + ti * E;
                    d.ldc = n2 * NB;
                    d.kt_begin = tj * EK;                   // Inv11^T is upper triangular: k >= row
                    d.kt_end = n1 * KPB;
                    d.alpha = 1.0;
                    descs.push_back(d);
                }
        }
        plan.stages.push_back({a0, (int)descs.size() - a0});
        // stage B: Inv21 (s2 x s1) = -Inv22 . T  and its transpose (s1 x s2) = -T^T . Inv22^T ;
        // Inv22 is lower triangular: k <= row tile of Inv22
        const int b0 = (int)descs.size();
        for (const Merge& m : merges) {
            if (m.level != lvl) continue;
            const SuperBlock& s = plan.sbs[m.sb];
            const int n1 = m.mid - m.lo, n2 = m.hi - m.mid;
            double* TT = tws + m.toff;
            for (int ti = 0; ti < n2 * SUB; ++ti)
                for (int tj = 0; tj < n1 * SUB; ++tj) {
                    GemmTileDesc d{};   // Inv21(ti,tj) = -sum_k Inv22[ti][k] T^T[tj][k]
                    d.P = s.inv + (size_t)(m.mid * NB + ti * E) * s.size + m.mid * NB;
                    d.ldp = s.size;
                    d.Q = TT + (size_t)(tj * E) * (n2 * NB);
                    d.ldq = n2 * NB;
                    d.C = s.inv + (size_t)(m.mid * NB + ti * E) * s.size + (m.lo * NB + tj * E);
                    d.ldc = s.size;
                    d.kt_begin = 0;
                    d.kt_end = (ti + 1) * EK;               // Inv22 is lower triangular: k <= row
                    d.alpha = -1.0;
                    descs.push_back(d);
                    GemmTileDesc t{};   // Inv21^T(tj,ti) = -sum_k T^T[tj][k] Inv22[ti][k]
                    t.P = TT + (size_t)(tj * E) * (n2 * NB);
                    t.ldp = n2 * NB;
                    t.Q = s.inv + (size_t)(m.mid * NB + ti * E) * s.size + m.mid * NB;
                    t.ldq = s.size;
                    t.C = s.invT + (size_t)(m.lo * NB + tj * E) * s.size + (m.mid * NB + ti * E);
                    t.ldc = s.size;
                    t.kt_begin = 0;
                    t.kt_end = (ti + 1) * EK;
                    t.alpha = -1.0;
                    descs.push_back(t);
                }
        }
        plan.stages.push_back({b0, (int)descs.size() - b0});
    }
    if ((e = hipMalloc((void**)&plan.descs_dev, (descs.size() + 1) * sizeof(GemmTileDesc))) != hipSuccess) return e;
    if (!descs.empty()) {
        e = hipMemcpyAsync(plan.descs_dev, descs.data(), descs.size() * sizeof(GemmTileDesc), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipStreamSynchronize(st);   // `descs` must outlive the copy
}

void factor_plan_destroy(FactorPlan& plan) {
    if (plan.descs_dev) (void)hipFree(plan.descs_dev);   // the inverse storage belongs to the arena
    plan.sbs.clear();
    plan.stages.clear();
    plan.descs_dev = nullptr;
    plan.tpart = nullptr;
    plan.mp = 0;
}

// ------------------------------------------------------------------------------------------------
// y[q][c] -= sum_s part[s][q][c]   (c < width): folds the row-split slabs of the transposed panel
// product into the right-hand side of the backward step, in slab order.
__global__ __launch_bounds__(256) void trsv_fold_kernel(double* __restrict__ y, long long ldy,
                                                        const double* __restrict__ part, int nsplit, int nrhs,
                                                        int width, BatchK bk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= width || batch_done(bk)) return;
    y = batch_ptr(y, bk); part = batch_ptr(part, bk);
    for (int q = 0; q < nrhs; ++q) {
        // eight slab values are requested before any is added (one memory round trip per eight instead of per
        // value); the additions stay in slab order
        double s = 0.0;
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[((long long)(sp + u) * nrhs + q) * width + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sp < nsplit; ++sp) s += part[((long long)sp * nrhs + q) * width + c];
        y[(long long)q * ldy + c] -= s;
    }
}

hipError_t launch_chol_solve(const double* L, int64_t ld, const FactorPlan& plan, int nrhs, double* R,
                             double* Y, hipStream_t st, const Batch& bt) {
    const int mp = plan.mp;
    hipError_t e;
    const int nsb = (int)plan.sbs.size();
    // forward: L y = r
    for (int k = 0; k < nsb; ++k) {
        const SuperBlock& s = plan.sbs[k];
        e = launch_gemv_n(s.inv, s.size, s.size, s.size, nrhs, R + s.row0, mp, nullptr, nullptr, Y + s.row0, mp, st, 1.0, bt);
        if (e != hipSuccess) return e;
        const int below = mp - (s.row0 + s.size);
        if (below > 0) {
            double* rb = R + s.row0 + s.size;
            e = launch_gemv_n(L + (size_t)(s.row0 + s.size) * ld + s.row0, ld, below, s.size, nrhs, Y + s.row0, mp, rb,
                              rb + mp, rb, mp, st, -1.0, bt);
            if (e != hipSuccess) return e;
        }
    }
    // backward: L^T v = y   (solution written over R)
    for (int k = nsb - 1; k >= 0; --k) {
        const SuperBlock& s = plan.sbs[k];
        const int below = mp - (s.row0 + s.size);
        if (below > 0) {
            e = launch_gemv_t(L + (size_t)(s.row0 + s.size) * ld + s.row0, ld, below, s.size, nrhs,
                              R + s.row0 + s.size, mp, plan.tpart, st, 0, bt);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(trsv_fold_kernel, dim3((s.size + 255) / 256, 1, bt.count), dim3(256), 0, st, Y + s.row0,
                               (long long)mp, plan.tpart, below / GEMVT_ROWS, nrhs, s.size, batch_k(bt));
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        e = launch_gemv_n(s.invT, s.size, s.size, s.size, nrhs, Y + s.row0, mp, nullptr, nullptr, R + s.row0, mp, st, 1.0, bt);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace lpipm
