// solver.hip -- the device-resident interior-point loop behind the C ABI (include/lpipm.h).
//
// Host side of InteriorPoint::solve_normal_form (interior_point/mod.rs:199-240): A is uploaded
// once; every iteration is a fixed sequence of kernel launches on the ctx's stream with all
// scalars on device; the host reads back one 96-byte status record per iteration to decide
// termination exactly as mod.rs:230-235 does.  There is no CPU fallback: every numerical step is
// a HIP kernel, and a missing/unusable device is an error.
//
// Algebraic reuse that leaves results identical to the reference (same inputs, same arithmetic):
//   * (p, q) = sym_solve(c, b) is computed once per iteration; the reference recomputes it for the
//     corrector with the same factor and inputs (feasible_point.rs:149 -> newton_equations.rs:187);
//   * r_P, r_D of the next get_delta (feasible_point.rs:122-123) are the vectors whose norms the
//     indicators just took (residual.rs:22-26) at the same point;
//   * the predictor's two sym_solve calls share one pass over A per GEMV and one 2-RHS solve.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <thread>
#include <atomic>
#include <chrono>
#include "vec_kernels.hpp"

using namespace lpipm;

namespace lpipm {
static thread_local std::string g_err_detail;
void set_error_detail(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s -> %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err_detail = buf;
}
}  // namespace lpipm

namespace lpipm { extern long long* g_diag_stamps; }
namespace lpipm {
const char* lp_knob(const char* name) {
    const char* master = getenv("LPIPM_EXPERIMENTAL");
    return (master && master[0] == '1') ? getenv(name) : nullptr;
}
}  // namespace lpipm
enum { T_VEC = 0, T_ADAT, T_POTRF, T_TRSV, T_GEMV, T_NTAGS };

struct lpipm_ctx {
    int device = 0;
    hipStream_t st = nullptr;
    int num_cu = 256;
    bool has_problem = false;
    uint64_t m = 0, n = 0;
    int mp = 0, np = 0, nblk = 1, nsplit = 1;
    int ns = 0, nx = 0, npa = 0;   // slack columns (not stored), structural columns, their padded count = lda of A
    // per-LP device state lives in one arena; a lockstep batch of B LPs has B of them, bstride bytes apart
    char* arena = nullptr;
    size_t arena_bytes = 0, bstride = 0;
    int B = 1;
    Batch bt;                    // what the solve path hands to every launcher (count, stride, done flags)
    Batch bt_head;               // same with the done test always on: the speculatively enqueued head of an iteration
    hipEvent_t ev_status = nullptr;   // recorded behind the status copy of an iteration
    std::vector<void*> kallocs;  // buffers of the stand-alone kernel entry points
    // problem + state + work
    FactorPlan plan, kplan;
    double *tau = nullptr, *ktau = nullptr;   // Householder scalars of the QR arms
    double *A = nullptr, *M = nullptr, *ws = nullptr, *Y = nullptr, *ATpart = nullptr, *xout = nullptr;
    double *M0 = nullptr, *R0 = nullptr, *Rho = nullptr, *symv_ws = nullptr;   // refinement of the Cholesky solve
    // A.D.A^T as (tile, chunk) units with an in-launch combine (launch_adat_units; kernels_gemm.hip)
    int units_env = 1;                   // LPIPM_ADAT_UNITS=0: the round-2 kernel (data-parallel tiles + stream-K + fix-up launch)
    bool units = false;                  // this problem runs the units kernel (geometry: the slabs fit the budget)
    bool grouped_reduce = false;         // column split over ranks: M is reduced group by group behind the running launch
    int cpt = 1, upc = 1;                // chunks per tile, chunks per unit
    int nunits = 0, nunits_grp = 0;
    int2* unit_list = nullptr;           // (tile, first chunk) in dispatch order: chunk-major over the XCD-aware tile order
    int2* unit_list_grp = nullptr;       // column-group-major (tile indices into tile_list_grp): groups complete one after the other
    unsigned int* tile_cnt = nullptr;    // arena: arrival counters of the tiles, then the group words (one memset clears both)
    unsigned int* grp_cnt = nullptr;
    size_t cnt_bytes = 0;
    bool cnt_dirty = true;               // the arrival words may be non-zero: the next plain units launch clears them first (a plain
                                         // launch leaves them zero itself; launches with group words do not)
    unsigned int* wait_timeout = nullptr;   // arena: set by a wait kernel that gave up (a producer that never ran)
    unsigned int* timeout_host = nullptr;   // pinned mirror, read with the status record
    // factorisation beside A.D.A^T (enqueue_factor_grouped): CU-masked streams, column groups of the tile list
    hipStream_t st_a = nullptr, st_b = nullptr, st_u = nullptr;
    double* ws_upd = nullptr;            // stream-K slabs of the left-looking updates (they run beside A.D.A^T: own buffer)
    size_t ws_upd_slabs = 0;
    unsigned int* sk_claim_upd = nullptr;
    hipEvent_t ev_adat_done = nullptr;
    int overlap_cus = 0;                 // CUs per XCC reserved for the chain stream (0: no masked streams)
    bool overlap = false;                // this problem can be factorised beside its A.D.A^T (geometry)
    bool factor_in_head = false;         // ... and the current solve does so (Cholesky arm, no column split, no graph replay)
    uint64_t overlap_sections = 0;       // profiling: sections enqueued in this solve
    std::vector<int> grp_off, grp_nt;    // tile sub-list of every column group (outer panel of the factorisation)
    hipEvent_t ev_fork = nullptr;
    PotrfLookahead la;                   // trailing updates of one factorisation beside the next panel's chain (launch_potrf)
    // a lockstep batch as two half-batches driven by two host threads on two streams (solve_lockstep): views of this
    // context that share its arena (every pointer is LP 0's; a view's launches cover the LPs [bt.first, bt.first + B))
    bool is_view = false;
    int halves_env = 1;                  // LPIPM_HALVES=0: one stream for the whole batch
    bool pred_done = false;              // the last residual launch also ran the next iteration's k_pred_setup
    std::vector<lpipm_ctx*> halves;
    std::vector<hipEvent_t> ev_ready, ev_chain, ev_adat;
    int refine = 0;              // set from the environment by lpipm_create.  0 (default): plain solves; LPIPM_REFINE=2: every
                                 //   solve of every iteration refined; =1: only from mu / mu_0 <= refine_below() on.
                                 //   Built because ~1 % of the C4 members took a poor last step (alpha 0.987 for 0.99995) and
                                 //   one iteration more than the oracle; measured on all 256 members, no mode removes such
                                 //   members -- each variant has its own one or two, always among the members whose last
                                 //   d_tau is ill-determined in fp64 (tests/golden/make_c4_members.py, margin()): the oracle
                                 //   does the same under a permutation of its columns.  Refinement costs 7-20 % and is off.
    bool refine_now = false;     // the iteration being enqueued refines its solves (host mirror of the LPs' skip_refine words)
    int2* tile_list = nullptr;
    int2* tile_list_grp = nullptr;      // the same tiles grouped by column group (behind tile_list in one allocation)
    size_t ws_slabs = 0;                // stream-K slabs (TILE x TILE doubles each) the A.D.A^T / update launches may need
    unsigned int* sk_claim = nullptr;   // claim word of the dynamic stream-K chunks of A.D.A^T
    int ntiles = 0, adat_nwg = 1;
    VecArgs va{};
    StatusRec* status_host = nullptr;  // pinned, status_cap records
    uint32_t seq_counter = 0;          // sequence numbers of the status records (VecArgs::status_seq)
    bool spin_status = false;          // this solve waits for an iteration by watching the records' sequence words (wait_status)
    double* x_pinned = nullptr;        // pinned bounce buffer of the solution (a D2H copy into the caller's pageable array takes
    size_t x_pinned_cap = 0;           //   the runtime's staged path: ~40 us more per solve than pinned + memcpy)
    size_t status_cap = 0;
    // stand-alone potrf/solve buffers
    double *kM = nullptr, *kM0 = nullptr, *kR = nullptr, *kY = nullptr;
    int32_t* kinfo = nullptr;
    int kmp = 0;
    bool kchol_valid = false;
    // profiling
    int profiling = 0;           // 0 off, 1 every phase, 2 only the A.D.A^T launches (2 events per iteration)
    std::vector<hipEvent_t> events;
    std::vector<int> mark_tags;
    size_t nmarks = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    double tag_ms[T_NTAGS] = {0, 0, 0, 0, 0};
    uint64_t gemv_passes = 0;
    lpipm_phase_times times{};
    // batch mode: extra contexts (own stream + buffers) driven by host threads, see lpipm_solve_batch
    std::vector<lpipm_ctx*> workers;
    int batch_concurrency = 0;   // 0 = auto
    int lockstep_max = -1;       // lpipm_solve_batch: -1 auto, 0 never group same-shape members, > 0 largest group
    // captured iteration (hipGraph): one executable graph per (ip, options) key, valid while the buffers live
    struct IterGraph { int ip; int refine; double alpha0, tol; hipGraphExec_t exec; };
    std::vector<IterGraph> graphs;
    int use_graph = -1;          // -1: decide from the environment at first use
    bool no_speculate = false;
    // n-split mode (one LP split by columns over ranks; BASELINE config C5): the collective is the caller's
    bool colsplit = false;
    int rank = 0, world = 1;
    lpipm_allreduce_fn coll = nullptr;
    void* coll_user = nullptr;
    bool coll_on_stream = false;  // the callback enqueues the reduction on the ctx's stream itself (no drain before the call)
    double* gs = nullptr;        // 8 doubles: sums / minima that must be reduced across ranks
    hipStream_t st_c = nullptr;  // communication stream: the column groups of M are packed, reduced and unpacked here, behind the
    hipEvent_t ev_c0 = nullptr, ev_c1 = nullptr;   //   group words of the A.D.A^T launch that is still running on the solver's stream
    double* mpack = nullptr;     // contiguous image of the lower block-triangle of M for its all-reduce
    size_t mpack_count = 0;
};

static void destroy_views(lpipm_ctx* c);      // half-batch views of a lockstep batch (solve_lockstep)
namespace lpipm { lpipm_ctx_device lpipm_ctx_device_of(lpipm_ctx* c) { return lpipm_ctx_device{c->device, c->st}; } }

// The factorisation beside A.D.A^T (enqueue_factor_grouped) unless LPIPM_OVERLAP says otherwise: see lpipm_create.
constexpr bool OVERLAP_DEFAULT = false;

static void drop_graphs(lpipm_ctx* c) {
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

// Cross-rank reduction of `count` doubles at a device pointer, ordered after everything enqueued on the ctx's stream
// so far.  Default contract: the stream is drained first and the callee returns when the result is in place.
// lpipm_set_collective_on_stream(ctx, 1): nothing is drained -- the callee enqueues the reduction ON the stream it is
// given (ncclAllReduce(..., stream)) and returns at once; stream order does the rest, and the M panels' reduction
// overlaps whatever the host enqueues next.
static int ctx_allreduce(lpipm_ctx* c, double* ptr, uint64_t count, int op, hipStream_t on = nullptr) {
    if (!c->colsplit || c->world <= 1) return LPIPM_OK;
    if (!c->coll) return LPIPM_ERR_BAD_ARGUMENT;
    hipStream_t st = on ? on : c->st;             // (the M groups are reduced on the communication stream, see enqueue_head)
    if (!c->coll_on_stream) LP_HIP(hipStreamSynchronize(st));
    if (c->coll(c->coll_user, ptr, count, op, (void*)st) != 0) {
        g_err_detail = "the all-reduce callback of lpipm_set_collective reported a failure";
        return LPIPM_ERR_HIP;
    }
    return LPIPM_OK;
}
static int xrank_fn(void* self, double* ptr, int count, int op) { return ctx_allreduce((lpipm_ctx*)self, ptr, (uint64_t)count, op); }

// ------------------------------------------------------------------------------------------------
static int free_list(std::vector<void*>& v) {
    for (void* p : v) (void)hipFree(p);
    v.clear();
    return 0;
}
template <typename T>
static int dalloc(std::vector<void*>& list, std::vector<size_t>* sizes, T** out, size_t count, hipStream_t st) {
    void* p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    LP_HIP(hipMalloc(&p, bytes));
    list.push_back(p);
    if (sizes) sizes->push_back(bytes);
    LP_HIP(hipMemsetAsync(p, 0, bytes, st));
    *out = (T*)p;
    return LPIPM_OK;
}
#define LP_TRY(expr) do { int rc__ = (expr); if (rc__ != LPIPM_OK) return rc__; } while (0)

static void prof_mark(lpipm_ctx* c, int tag, bool adat_bracket = false) {
    if (!c->profiling || (c->profiling == 2 && !adat_bracket)) return;
    if (c->nmarks == c->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->events.push_back(e);
        c->mark_tags.push_back(0);
    }
    c->mark_tags[c->nmarks] = tag;
    (void)hipEventRecord(c->events[c->nmarks], c->st);
    ++c->nmarks;
}
// Adds up the intervals between the first `upto` marks (all of them by default); call when those events have
// completed.  Later marks (the speculatively enqueued head of the next iteration) move to the front.
static void prof_collect(lpipm_ctx* c, size_t upto = (size_t)-1) {
    if (!c->profiling) return;
    if (upto > c->nmarks) upto = c->nmarks;
    for (size_t i = 1; i < upto; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->events[i - 1], c->events[i]) == hipSuccess)
            c->tag_ms[c->mark_tags[i]] += ms;
    }
    for (size_t i = upto; i < c->nmarks; ++i) {
        std::swap(c->events[i - upto], c->events[i]);
        c->mark_tags[i - upto] = c->mark_tags[i];
    }
    c->nmarks -= upto;
}

// A.D.A^T time of one completed side-by-side section (its events have completed): the one launch on the throughput stream
static void prof_collect_overlap(lpipm_ctx* c, uint64_t section) {
    if (!c->profiling || !c->factor_in_head) return;
    hipEvent_t* ev = c->ev_adat.data() + (section & 1) * 2;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) c->tag_ms[T_ADAT] += ms;
}

// ------------------------------------------------------------------------------------------------
extern "C" void lpipm_default_opts(lpipm_opts* o) {  // interior_point/mod.rs:50-60
    if (!o) return;
    o->tol = 1e-8; o->alpha0 = 0.99995; o->max_iter = 1000; o->ip = 1;
    o->solver_type = LPIPM_SOLVER_CHOLESKY; o->disp = 0;
}

extern "C" const char* lpipm_strerror(int s) {  // error.rs:10-28
    switch (s) {
        case LPIPM_OK: return "Ok";
        case LPIPM_UNCONSTRAINED:
            return "The problem is unconstrained, meaning the solution is the all-zeros vector if `c` is nonnegative, or unbounded otherwise.";
        case LPIPM_NUMERICAL_PROBLEM:
            return "The solver encountered numerical problems it could not recover from. Likely causes are linearly dependent constraints or variables whose scale differs by multiple orders of magnitude.";
        case LPIPM_INVALID_PARAMETER: return "A parameter was set to an invalid value";
        case LPIPM_INCOMPATIBLE_DIMENSIONS: return "The dimensions of your cost- and constraint arrays do not align.";
        case LPIPM_INFEASIBLE: return "The solver finished successfully, it appears that the problem is infeasible.";
        case LPIPM_UNBOUNDED: return "The solver finished successfully, it appears that your problem is unbounded.";
        case LPIPM_ITERATION_LIMIT:
            return "The solver failed to converge within the maximum number of iterations.";
        case LPIPM_ERR_HIP: return "HIP runtime error (see lpipm_last_error_detail)";
        case LPIPM_ERR_NO_PROBLEM: return "no problem uploaded on this context";
        case LPIPM_ERR_UNSUPPORTED: return "not supported by the HIP backend yet";
        case LPIPM_ERR_BAD_ARGUMENT: return "bad argument";
        default: return "unknown status";
    }
}
extern "C" const char* lpipm_last_error_detail(void) { return g_err_detail.c_str(); }

extern "C" int lpipm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// linear_program.rs:125-169
extern "C" int lpipm_problem_build(uint64_t n, uint64_t m_ub, const double* A_ub, const double* b_ub,
                                   uint64_t m_eq, const double* A_eq, const double* b_eq, const double* c,
                                   double* A_out, double* b_out, double* c_out, uint64_t* n_slack_out) {
    if (m_ub + m_eq == 0) return LPIPM_UNCONSTRAINED;                       // :134-136
    if (!c || !A_out || !b_out || !c_out || !n_slack_out) return LPIPM_ERR_BAD_ARGUMENT;
    if ((m_ub && (!A_ub || !b_ub)) || (m_eq && (!A_eq || !b_eq))) return LPIPM_ERR_BAD_ARGUMENT;
    const uint64_t m = m_ub + m_eq, ns = n + m_ub;
    for (uint64_t i = 0; i < m; ++i) {                                       // :145-156
        const double* src = i < m_ub ? A_ub + i * n : A_eq + (i - m_ub) * n;
        double* dst = A_out + i * ns;
        for (uint64_t j = 0; j < n; ++j) dst[j] = src[j];
        for (uint64_t j = 0; j < m_ub; ++j) dst[n + j] = (i == j) ? 1.0 : 0.0;
    }
    for (uint64_t i = 0; i < m; ++i) b_out[i] = i < m_ub ? b_ub[i] : b_eq[i - m_ub];  // :157-158
    for (uint64_t j = 0; j < ns; ++j) c_out[j] = j < n ? c[j] : 0.0;                  // :159-160
    *n_slack_out = m_ub;                                                               // :161
    return LPIPM_OK;
}

extern "C" int lpipm_create(int device, lpipm_ctx** out) {
    if (!out) return LPIPM_ERR_BAD_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    LP_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        g_err_detail = "device index out of range (no usable HIP device?)";
        return LPIPM_ERR_HIP;
    }
    LP_HIP(hipSetDevice(device));
    lpipm_ctx* c = new lpipm_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&c->status_host, sizeof(StatusRec), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipEventCreate(&c->ev_begin) != hipSuccess || hipEventCreate(&c->ev_end) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_status, hipEventDisableTiming) != hipSuccess) {
        g_err_detail = "failed to create stream / pinned status / events";
        delete c;
        return LPIPM_ERR_HIP;
    }
    c->status_cap = 1;
    // LPIPM_REFINE (see lpipm_ctx::refine) is read ONCE, here: the arena layout depends on it (M0, R0, Rho and the symv slabs
    // exist only for a refining context: 134 MB at C3, 2 GB at m = 16384, per member of a lockstep batch)
    { const char* e = lp_knob("LPIPM_REFINE"); c->refine = !e ? 0 : (e[0] == '2' ? 2 : (e[0] == '1' ? 1 : 0)); }
    // LPIPM_ADAT_UNITS: 0 = the round-2 kernel everywhere, 2 = the units kernel for single LPs too (measurement / test knob);
    // default 1 = units kernel for lockstep batches and for the side-by-side factorisation, round-2 kernel for a single LP
    // (measured per launch, units vs round-2: 512x1024 0.042 / 0.045 ms, 1024x2048 0.102 / 0.097, 2048x4096 0.469 / 0.458,
    // 4096x8192 2.47 / 2.39 standalone and 2.32 / 2.25 inside a solve; C4 lockstep shard 1732 vs 1674 LP/s)
    { const char* e = lp_knob("LPIPM_ADAT_UNITS"); c->units_env = !e ? 1 : (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)); }
    { const char* e = lp_knob("LPIPM_HALVES"); c->halves_env = (e && e[0] == '0') ? 0 : 1; }
    if (hipHostMalloc((void**)&c->timeout_host, sizeof(unsigned int)) != hipSuccess) {
        g_err_detail = "failed to allocate the pinned time-out word";
        lpipm_destroy(c);
        return LPIPM_ERR_HIP;
    }
    *c->timeout_host = 0;
    std::memset(c->status_host, 0, sizeof(StatusRec));
    // CU-masked streams for the factorisation that runs beside A.D.A^T (enqueue_factor_grouped).  Mask bit i is CU i/8 of
    // XCC i%8 (scripts/diag/cu_mask_probe.cpp; an XCC with no bit set would be unrestricted): the chain stream (st_b) gets
    // CUs 0..R-1 of every XCC -- a diagonal-block kernel needs a whole CU's LDS, and on a chip full of A.D.A^T workgroups it
    // would wait for one --, the throughput streams (st_a: the one A.D.A^T launch; st_u: the left-looking updates that run
    // beside it) the rest.  LPIPM_OVERLAP=0 switches the scheme off, LPIPM_OVERLAP_CUS=R sets R (default 4).  Only on the
    // 8 x 32 CU layout it was measured on.
    {
        const char* on = lp_knob("LPIPM_OVERLAP");
        int R = 4;
        if (const char* e = lp_knob("LPIPM_OVERLAP_CUS")) { const int v = atoi(e); if (v >= 1 && v <= 16) R = v; }
        if (OVERLAP_DEFAULT ? !(on && on[0] == '0') : (on && on[0] == '1')) if (c->num_cu == 256 && c->units_env) {
            uint32_t ma[8], mb[8];
            for (int w = 0; w < 8; ++w) { ma[w] = 0; mb[w] = 0; }
            for (int i = 0; i < 256; ++i) ((i / 8) < R ? mb : ma)[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&c->st_a, 8, ma) == hipSuccess &&
                hipExtStreamCreateWithCUMask(&c->st_u, 8, ma) == hipSuccess &&
                hipExtStreamCreateWithCUMask(&c->st_b, 8, mb) == hipSuccess &&
                hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&c->ev_adat_done, hipEventDisableTiming) == hipSuccess) {
                c->overlap_cus = R;
            } else {
                (void)hipGetLastError();
                if (c->st_a) (void)hipStreamDestroy(c->st_a);
                if (c->st_u) (void)hipStreamDestroy(c->st_u);
                if (c->st_b) (void)hipStreamDestroy(c->st_b);
                c->st_a = c->st_b = c->st_u = nullptr;
            }
        }
    }
    // Side stream for the look-ahead of the factorisation's trailing updates (launch_potrf; used from m = 4096, see there for
    // the measurements; LPIPM_LOOKAHEAD=0 switches it off, =1 lowers the threshold to m = 1536).  CU-masked (bit i = CU i/8 of XCC i%8): the first R CUs of every XCC stay free for the chain
    // stream's kernels -- the diagonal-block kernel needs a CU to itself (150 KB of LDS) and would otherwise wait for a
    // side-stream tile to drain.  LPIPM_LOOKAHEAD_CUS=R sets R (default 8; 0: an unmasked low-priority stream).
    {
        const char* on = lp_knob("LPIPM_LOOKAHEAD");
        int R = 8;
        if (const char* e = lp_knob("LPIPM_LOOKAHEAD_CUS")) { const int v = atoi(e); if (v >= 0 && v <= 16) R = v; }
        if (on && on[0] == '1') c->la.min_nb = 3 * POTRF_OUTER;
        if (!(on && on[0] == '0') && c->num_cu == 256) {
            uint32_t mk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 256; ++i) if ((i / 8) >= R) mk[i / 32] |= 1u << (i % 32);
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = least priority (numerically largest)
            hipError_t e = R > 0 ? hipExtStreamCreateWithCUMask(&c->la.side, 8, mk) : hipStreamCreateWithPriority(&c->la.side, hipStreamNonBlocking, lo);
            constexpr int NEV = 64;              // outer panels of the largest factorisation (m <= 32768)
            for (int i = 0; i < 2 * NEV && e == hipSuccess; ++i) {
                hipEvent_t ev = nullptr;
                e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                if (e == hipSuccess) (i < NEV ? c->la.ev_chain : c->la.ev_rest).push_back(ev);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (hipEvent_t ev : c->la.ev_chain) (void)hipEventDestroy(ev);
                for (hipEvent_t ev : c->la.ev_rest) (void)hipEventDestroy(ev);
                c->la.ev_chain.clear(); c->la.ev_rest.clear();
                if (c->la.side) (void)hipStreamDestroy(c->la.side);
                c->la.side = nullptr;
            }
        }
    }
    *out = c;
    return LPIPM_OK;
}

extern "C" void lpipm_destroy(lpipm_ctx* c) {
    if (!c) return;
    for (lpipm_ctx* w : c->workers) lpipm_destroy(w);
    c->workers.clear();
    destroy_views(c);
    (void)hipSetDevice(c->device);
    if (c->st) (void)hipStreamSynchronize(c->st);
    drop_graphs(c);
    if (c->arena) (void)hipFree(c->arena);
    if (c->tile_list) (void)hipFree(c->tile_list);
    free_list(c->kallocs);
    if (c->mpack) (void)hipFree(c->mpack);
    if (c->st_c) { (void)hipStreamSynchronize(c->st_c); (void)hipStreamDestroy(c->st_c); }
    if (c->ev_c0) (void)hipEventDestroy(c->ev_c0);
    if (c->ev_c1) (void)hipEventDestroy(c->ev_c1);
    factor_plan_destroy(c->plan);
    factor_plan_destroy(c->kplan);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->ev_status) (void)hipEventDestroy(c->ev_status);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipEvent_t e : c->ev_ready) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_chain) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_adat) (void)hipEventDestroy(e);
    if (c->la.side) { (void)hipStreamSynchronize(c->la.side); (void)hipStreamDestroy(c->la.side); }
    for (hipEvent_t e : c->la.ev_chain) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->la.ev_rest) (void)hipEventDestroy(e);
    if (c->st_a) { (void)hipStreamSynchronize(c->st_a); (void)hipStreamDestroy(c->st_a); }
    if (c->st_u) { (void)hipStreamSynchronize(c->st_u); (void)hipStreamDestroy(c->st_u); }
    if (c->st_b) { (void)hipStreamSynchronize(c->st_b); (void)hipStreamDestroy(c->st_b); }
    if (c->ev_adat_done) (void)hipEventDestroy(c->ev_adat_done);
    if (c->timeout_host) (void)hipHostFree(c->timeout_host);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->x_pinned) (void)hipHostFree(c->x_pinned);
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
}

// Order in which the lower-triangular 128x128 tiles of M are handed to workgroups.  Workgroups are
// renumbered so that 64 consecutive tiles run on one XCD (one L2): full off-diagonal 8x8 super-blocks
// come first, each exactly one such chunk (16 row panels of A feed 64 tiles); the triangular
// diagonal super-blocks (36 tiles each) follow and are the ones that straddle chunk boundaries.
// The same tiles ordered for the factorisation that runs beside A.D.A^T: column group g (tile columns 4g .. 4g+3, one
// outer panel of the factorisation) is one contiguous sub-list; inside it row by row, so that consecutive stream-K
// claims of one k-range share a row panel of A.
static std::vector<int2> adat_tile_order_grouped(int nt, std::vector<int>& off, std::vector<int>& cnt) {
    std::vector<int2> v;
    off.clear(); cnt.clear();
    for (int g = 0; g * POTRF_OUTER < nt; ++g) {
        off.push_back((int)v.size());
        const int c0 = g * POTRF_OUTER, c1 = c0 + POTRF_OUTER < nt ? c0 + POTRF_OUTER : nt;
        for (int ti = c0; ti < nt; ++ti)
            for (int tj = c0; tj < c1 && tj <= ti; ++tj) v.push_back(make_int2(ti, tj));
        cnt.push_back((int)v.size() - off.back());
    }
    return v;
}
static std::vector<int2> adat_tile_order(int nt) {
    std::vector<int2> v;
    v.reserve((size_t)nt * (nt + 1) / 2);
    const int ns = (nt + 7) / 8;
    auto emit = [&](int SI, int SJ) {
        for (int ti = SI * 8; ti < nt && ti < SI * 8 + 8; ++ti)
            for (int tj = SJ * 8; tj < SJ * 8 + 8 && tj <= ti; ++tj) v.push_back(make_int2(ti, tj));
    };
    for (int SI = 0; SI < ns; ++SI)
        for (int SJ = 0; SJ < SI; ++SJ) emit(SI, SJ);
    for (int SI = 0; SI < ns; ++SI) emit(SI, SI);
    return v;
}

// Output tile edge of the inverse-merge GEMMs: 64 (a stage of one LP is a few dozen latency-bound tiles: factorisation
// 422 -> 358 us at m = 512, 719 -> 547 at 1024, 2556 -> 2398 at 4096; a batch that fills the chip is indifferent).
// LPIPM_MERGE_EDGE=128 restores the 128x128 tiles (measurement knob, scripts/potrf_sizes.py).
// Width of the diagonal super-blocks whose explicit inverses feed the triangular solves: a function of the problem
// size alone, so that an LP goes through exactly the same arithmetic alone and as a member of a lockstep batch (the
// two paths are bit-identical, tests/test_gpu_c4_members.py).  512 up to m = 2048: the last doubling level of the
// inverse (1024) costs a flop-bound batch more than the two solve steps it saves, and a single small LP about as much
// as it gains.  LPIPM_SUPER=<multiple of 128> overrides it (measurement knob).
static int super_for(int mp) {
    if (const char* e = lp_knob("LPIPM_SUPER")) { const int w = atoi(e); if (w >= NB && w % NB == 0 && w <= 4096) return w; }
    return mp <= 2048 ? 512 : SUPER;
}
static int merge_edge_for(int) {
    // 32x32 merge tiles: a stage is one round of tiles on the chain (factorisation m = 512: 194 -> 177 us, 1024: 364 -> 346,
    // 4096: 1865 -> 1854; the lockstep C4 batch is indifferent: 1745 LP/s either way)
    if (const char* e = lp_knob("LPIPM_MERGE_EDGE")) { const int v = atoi(e); return (v == 128 || v == 64) ? v : 32; }
    return 32;
}

// The unit list of a single LP's A.D.A^T launch, dealt to the XCDs.  Workgroup b of a launch runs on XCD b % 8 (round-robin
// dispatch), so entry b of the list belongs to XCD b % 8: every XCD gets its OWN tiles (full rounds of 512 tiles: 64
// consecutive tiles of the order = one 8 x 8 super-block sharing 16 row panels of A; the rest in contiguous eighths) and
// walks them chunk by chunk -- the workgroups resident on one XCD (one L2) are one k-range of neighbouring tiles for the
// whole launch, like the data-parallel phase of the round-2 kernel.  Shorter lists are padded with no-op entries.
// tiles: indices into the launch's tile list, in its order; chunks q0, q0 + upc, ... < cpt per tile.
static void deal_units(const std::vector<int>& tiles, int cpt, int upc, std::vector<int2>& out) {
    std::vector<int> own[8];
    const int nt = (int)tiles.size(), full = nt / 512 * 512, rest = nt - full;
    for (int i = 0; i < full; ++i) own[(i % 512) / 64].push_back(tiles[(size_t)i]);
    for (int x = 0; x < 8; ++x)
        for (int i = full + (int)((long long)rest * x / 8); i < full + (int)((long long)rest * (x + 1) / 8); ++i) own[x].push_back(tiles[(size_t)i]);
    size_t longest = 0;
    for (int x = 0; x < 8; ++x) longest = own[x].size() > longest ? own[x].size() : longest;
    const int nq = (cpt + upc - 1) / upc;
    for (int q = 0; q < nq; ++q)                           // chunk-major inside an XCD's list
        for (size_t i = 0; i < longest; ++i)
            for (int x = 0; x < 8; ++x)
                out.push_back(i < own[x].size() ? make_int2(own[x][i], q * upc) : make_int2(-1, 0));
}

// How the A.D.A^T launch of this geometry is cut up (a function of mp, npa, the batch count and the CU count alone).
static void plan_adat(lpipm_ctx* c, int count) {
    const int nt = c->mp / TILE;
    c->ntiles = nt * (nt + 1) / 2;
    // workgroups per LP of the round-2 A.D.A^T launch (LPIPM_ADAT_UNITS=0, and contractions whose slabs would not fit):
    // stream-K over the chip's share of one LP; a batch that fills the chip with whole tiles needs no k-split
    if (count == 1) c->adat_nwg = gemm_streamk_nwg(c->ntiles, c->npa / BK, c->num_cu);
    else if ((long long)count * c->ntiles >= 2LL * c->num_cu) {
        // more tiles than resident workgroups: each LP gets its share of the 2*CUs slots and stream-K
        // balances its tiles over them (no tail round of a few leftover tiles)
        c->adat_nwg = 2 * c->num_cu / count;
        if (c->adat_nwg < 1) c->adat_nwg = 1;
        if (c->adat_nwg > c->ntiles) c->adat_nwg = c->ntiles;
    } else {
        c->adat_nwg = gemm_streamk_nwg(c->ntiles, c->npa / BK, c->num_cu / count);
        if (c->adat_nwg < c->ntiles) c->adat_nwg = c->ntiles;
    }
    c->ws_slabs = gemm_streamk_slabs(c->ntiles, c->npa / BK, c->adat_nwg);
    // A.D.A^T as (tile, chunk) units: every chunk sum goes through its own slab (ntiles x cpt slabs of 128 KiB per LP:
    // 0.55 GB at C3, 38 MB per member at C4) -- up to 4 GiB per LP, beyond that (m = 16384: 34 GB) the round-2 kernel
    c->cpt = adat_units_cpt(c->npa);
    c->units = c->units_env != 0 && (size_t)c->ntiles * c->cpt * TILE * TILE * sizeof(double) <= ((size_t)4 << 30) &&
               (count > 1 || c->units_env == 2 || c->st_a != nullptr || c->cpt == 1 || c->ntiles <= 16 || c->ntiles * c->cpt >= 256);
    // (tiny single LPs -- up to 16 tiles -- : one launch and one memset less, 0.042 vs 0.045 ms at 512x1024;
    //  a single LP with few tiles AND several chunks -- 1000x5000: 36 tiles x 3 -- keeps the round-2 kernel: one workgroup per
    //  tile adding the slabs at the end of a launch that never filled the chip costs more than the 16-way fix-up launch,
    //  0.196 vs 0.151 ms; everywhere else the units kernel is level or ahead -- 4096x8192 2.206 vs 2.22 ms inside a solve,
    //  2048x16384 1.30 vs 1.60 -- carries no spill and leaves out the blocks above the diagonal of the diagonal tiles)
    // one LP split by columns over ranks: the units kernel signals M's column groups one by one, and each group's cross-rank
    // sum runs behind the rest of the launch (enqueue_head); its slabs may take up to 32 GiB there (C5: 17 GB per rank)
    if (count == 1 && c->world > 1 && c->units_env != 0 && nt <= 64 * POTRF_OUTER &&
        (size_t)c->ntiles * c->cpt * TILE * TILE * sizeof(double) <= ((size_t)32 << 30)) c->units = true;
    // a single LP: one chunk per unit (parallelism, and column groups that complete while the launch runs); a lockstep
    // batch: two chunks per unit -- whole tiles (one unit = all chunks, its own workgroup adds its slabs) leave the last of
    // 2.25 rounds of tiles a quarter full (C4 shard: 1633 LP/s, against 1706 with one chunk per unit, 1533 / 1521 / 1521 at
    // 2 / 1 / 4 chunks on a slower box)
    c->upc = count == 1 ? 1 : (c->cpt < 2 ? c->cpt : 2);
    { int kc, nbig, ks; if (adat_units_chunking(c->npa, &kc, &nbig, &ks) != nbig) c->upc = 1; }   // non-uniform chunks: one per unit
    if (c->units && c->ws_slabs < (size_t)c->ntiles * c->cpt) c->ws_slabs = (size_t)c->ntiles * c->cpt;
}

// Per-LP device state: one pass over a measuring arena sizes it, a second pass over the real one places it.
// Every LP of a lockstep batch gets the same layout, `bstride` bytes after the previous LP's.
static void bind_status_pinned(lpipm_ctx* c, bool allow);
static int layout_problem(lpipm_ctx* c, Arena& ar, bool build) {
    VecArgs& v = c->va;
    const size_t mp = (size_t)c->mp, np = (size_t)c->np;
    c->A = ar.take<double>(mp * c->npa);
    v.b = ar.take<double>(mp); v.c = ar.take<double>(np);
    v.x = ar.take<double>(np); v.y = ar.take<double>(mp); v.z = ar.take<double>(np);
    v.dinv = ar.take<double>(np); v.xs = ar.take<double>(np); v.r1 = ar.take<double>(np); v.rD = ar.take<double>(np);
    v.p = ar.take<double>(np); v.u = ar.take<double>(np); v.dx = ar.take<double>(np); v.dz = ar.take<double>(np);
    v.dxdz = ar.take<double>(np);
    v.rP = ar.take<double>(mp); v.rP2 = ar.take<double>(mp); v.q = ar.take<double>(mp); v.dy = ar.take<double>(mp);
    // chunk slabs of A.x: sized by the count the launches use (the STORED columns npa -- gemv_dual_chunks is not monotone:
    // 256-column chunks below 4096 columns, 1024-column chunks from there on, so np's count can be the smaller one)
    {
        const int ch_a = gemv_dual_chunks(c->npa), ch_n = gemv_dual_chunks((int)np);
        v.Ax = ar.take<double>(mp * (size_t)(ch_a > ch_n ? ch_a : ch_n));
    }
    v.W = ar.take<double>(2 * np); v.R = ar.take<double>(2 * mp);
    c->Y = ar.take<double>(2 * mp);
    c->ATpart = ar.take<double>((size_t)c->nsplit * 2 * np);
    v.ATpart = c->ATpart;
    v.S = ar.take<double>(64); v.red = ar.take<double>((size_t)RED_SLOTS * RED_STRIDE);
    v.status = ar.take<StatusRec>(1);
    v.potrf_info = ar.take<int32_t>(1); v.flags = ar.take<int>(1); v.done = ar.take<int>(1);
    v.skip_refine = ar.take<int>(1);
    c->M = ar.take<double>(mp * mp);
    LP_HIP(factor_plan_create(c->plan, c->M, c->mp, c->mp, ar, build, c->st, super_for(c->mp), merge_edge_for(c->B)));
    c->M0 = c->R0 = c->Rho = c->symv_ws = nullptr;
    if (c->refine > 0) {     // only the refined solves read the matrix itself
        c->M0 = ar.take<double>(mp * mp);
        c->R0 = ar.take<double>(2 * mp); c->Rho = ar.take<double>(2 * mp);
        c->symv_ws = ar.take<double>(symv_slab_doubles(c->mp));
    }
    c->tau = ar.take<double>(mp);
    c->gs = ar.take<double>(8);
    c->xout = ar.take<double>(np);
    c->sk_claim = ar.take<unsigned int>(1);
    c->sk_claim_upd = ar.take<unsigned int>(1);
    // arrival counters of the units kernel: one word per tile, then one per column group; cleared by ONE memset per launch
    // (a block of its own, a multiple of 16 bytes)
    c->cnt_bytes = (size_t)round_up(((size_t)c->ntiles + 64) * sizeof(unsigned int), 16);
    c->tile_cnt = (unsigned int*)ar.take<uint4>(c->cnt_bytes / 16);
    c->grp_cnt = c->tile_cnt + c->ntiles;
    c->wait_timeout = ar.take<unsigned int>(4);
    // chunk slabs of A.D.A^T (units kernel: every chunk of every tile; round-2 kernel: the stream-K remainder tiles), and
    // those of the left-looking updates that run beside it
    c->ws = ar.take<double>(c->ws_slabs * TILE * TILE);
    c->ws_upd = ar.take<double>(c->ws_upd_slabs * TILE * TILE);
    return LPIPM_OK;
}

// count LPs of one geometry (count == 1: the ordinary upload).  A/b/cc/c0: one entry per LP.
// `parts` (count == 1 only): the rows come as two blocks of nx = n - n_slack columns -- m_ub rows of A_ub, then
// m - m_ub rows of A_eq -- with b split the same way and c holding only the nx structural costs; the slack
// structure is then true by construction (lpipm_upload_ub_eq).
struct UploadParts { uint64_t m_ub; const double* A_ub; uint64_t lda_ub; const double* b_ub;
                     const double* A_eq; uint64_t lda_eq; const double* b_eq; };
static int upload_impl(lpipm_ctx* c, int count, uint64_t m, uint64_t n, const double* const* A, uint64_t lda,
                       const double* const* b, const double* const* cc, const double* c0, uint64_t n_slack,
                       const UploadParts* parts = nullptr) {
    if (!c || count < 1 || !cc) return LPIPM_ERR_BAD_ARGUMENT;
    if (!parts && (!A || !b || lda < n)) return LPIPM_ERR_BAD_ARGUMENT;
    for (int i = 0; i < count; ++i)
        if (!cc[i] || (!parts && (!A[i] || !b[i]))) return LPIPM_ERR_BAD_ARGUMENT;
    if (m == 0) return LPIPM_UNCONSTRAINED;  // linear_program.rs:134-136
    if (n == 0 || m > (1u << 20) || n > (1u << 24) || n_slack > n || n_slack > m) return LPIPM_ERR_BAD_ARGUMENT;
    // The hint is only used if the last n_slack columns really are [I; 0] (ProblemBuilder::build
    // guarantees it, linear_program.rs:147-156); anything else is treated as a dense matrix.
    if (!parts && (n_slack == n || count > 1)) n_slack = 0;
    for (uint64_t i = 0; i < m && n_slack && !parts; ++i) {
        const double* row = A[0] + i * lda + (n - n_slack);
        for (uint64_t j = 0; j < n_slack; ++j)
            if (row[j] != ((i == j) ? 1.0 : 0.0)) { n_slack = 0; break; }
    }
    LP_HIP(hipSetDevice(c->device));
    drop_graphs(c);   // kernel arguments depend on m, n, n_slack and the buffers
    destroy_views(c); // half-batch views copy the geometry and the buffers
    const uint64_t nx = n - n_slack;
    const int mp = (int)round_up(m, NB), np = (int)round_up(n, BK), npa = (int)round_up(nx, BK);
    hipStream_t st = c->st;
    if (!c->has_problem || mp != c->mp || np != c->np || npa != c->npa || count != c->B) {
        LP_HIP(hipStreamSynchronize(st));
        if (c->arena) { LP_HIP(hipFree(c->arena)); c->arena = nullptr; }
        if (c->tile_list) { LP_HIP(hipFree(c->tile_list)); c->tile_list = nullptr; }
        factor_plan_destroy(c->plan);
        c->has_problem = false;
        c->mp = mp; c->np = np; c->npa = npa; c->B = count;
        c->nsplit = mp / GEMVT_ROWS;
        const uint64_t big = m > n ? m : n;
        c->nblk = (int)((big + 255) / 256);
        if (c->nblk > RED_STRIDE) c->nblk = RED_STRIDE;
        const int nt = mp / TILE;
        std::vector<int2> order = adat_tile_order(nt);
        plan_adat(c, count);
        std::vector<int2> units, units_grp;
        if (c->units) {
            if (count == 1) {
                std::vector<int> all((size_t)c->ntiles);
                for (int t = 0; t < c->ntiles; ++t) all[(size_t)t] = t;
                deal_units(all, c->cpt, c->upc, units);
            } else {                                               // a batch: an LP's units all run on one XCD (xcd-major grid)
                for (int q = 0; q < c->cpt; q += c->upc)
                    for (int t = 0; t < c->ntiles; ++t) units.push_back(make_int2(t, q));
            }
        }
        c->nunits = (int)units.size();
        // Factorisation beside A.D.A^T: single LP, big enough that A.D.A^T can hide the factorisation's chain
        c->overlap = count == 1 && c->st_a != nullptr && mp >= 2048 && c->units && c->cpt > 1 && nt <= 64 * POTRF_OUTER;
        const bool grouped_reduce = count == 1 && c->world > 1 && c->units && nt <= 64 * POTRF_OUTER;
        std::vector<int2> grouped;
        c->ws_upd_slabs = 0;
        if (grouped_reduce && !c->overlap) {       // column-group-major unit list for the pipelined reduction of M (enqueue_head)
            grouped = adat_tile_order_grouped(nt, c->grp_off, c->grp_nt);
            for (size_t g = 0; g < c->grp_nt.size(); ++g) {
                std::vector<int> grp((size_t)c->grp_nt[g]);
                for (int t = 0; t < c->grp_nt[g]; ++t) grp[(size_t)t] = c->grp_off[g] + t;
                deal_units(grp, c->cpt, 1, units_grp);
            }
        }
        if (c->overlap) {
            grouped = adat_tile_order_grouped(nt, c->grp_off, c->grp_nt);
            const int wg_cus = c->num_cu - 8 * c->overlap_cus;
            for (size_t g = 0; g < c->grp_nt.size(); ++g) {
                std::vector<int> grp((size_t)c->grp_nt[g]);        // group-major; inside a group dealt to the XCDs, chunk-major
                for (int t = 0; t < c->grp_nt[g]; ++t) grp[(size_t)t] = c->grp_off[g] + t;
                deal_units(grp, c->cpt, 1, units_grp);
                const int ku = (int)g * POTRF_OUTER * NB / BK;      // contraction of the left-looking update of group g
                const size_t s2 = ku ? gemm_streamk_slabs(c->grp_nt[g], ku, gemm_streamk_nwg(c->grp_nt[g], ku, wg_cus)) : 0;
                if (s2 > c->ws_upd_slabs) c->ws_upd_slabs = s2;
            }
            while (c->ev_ready.size() < c->grp_nt.size()) {
                hipEvent_t e1, e2;
                LP_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
                LP_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
                c->ev_ready.push_back(e1); c->ev_chain.push_back(e2);
            }
            while (c->ev_adat.size() < 4) {     // {begin, end} x 2: the head of iteration k+1 is enqueued before iteration k's times are read
                hipEvent_t e3;
                LP_HIP(hipEventCreate(&e3));
                c->ev_adat.push_back(e3);
            }
        }
        c->nunits_grp = (int)units_grp.size();
        c->grouped_reduce = grouped_reduce && c->nunits_grp > 0;
        Arena measure;
        LP_TRY(layout_problem(c, measure, false));
        c->bstride = round_up(measure.off, 4096);
        c->arena_bytes = c->bstride * (size_t)count;
        LP_HIP(hipMalloc((void**)&c->arena, c->arena_bytes));
        LP_HIP(hipMemsetAsync(c->arena, 0, c->arena_bytes, st));
        Arena real;
        real.base = c->arena;
        LP_TRY(layout_problem(c, real, true));
        LP_HIP(hipMalloc((void**)&c->tile_list, (order.size() + grouped.size() + units.size() + units_grp.size() + 1) * sizeof(int2)));
        LP_HIP(hipMemcpyAsync(c->tile_list, order.data(), order.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        c->tile_list_grp = c->tile_list + order.size();
        c->unit_list = c->tile_list_grp + grouped.size();
        c->unit_list_grp = c->unit_list + units.size();
        if (!grouped.empty())
            LP_HIP(hipMemcpyAsync(c->tile_list_grp, grouped.data(), grouped.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        if (!units.empty())
            LP_HIP(hipMemcpyAsync(c->unit_list, units.data(), units.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        if (!units_grp.empty())
            LP_HIP(hipMemcpyAsync(c->unit_list_grp, units_grp.data(), units_grp.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        LP_HIP(hipStreamSynchronize(st));  // the lists must outlive the copies
        if ((size_t)count > c->status_cap) {
            if (c->status_host) (void)hipHostFree(c->status_host);
            c->status_host = nullptr; c->status_cap = 0;
            LP_HIP(hipHostMalloc((void**)&c->status_host, (size_t)count * sizeof(StatusRec), hipHostMallocCoherent | hipHostMallocMapped));
            std::memset(c->status_host, 0, (size_t)count * sizeof(StatusRec));
            c->status_cap = (size_t)count;
        }
        VecArgs& v = c->va;
        v.np = np; v.mp = mp; v.nblk = c->nblk; v.nsplit = c->nsplit;
        v.bcount = count; v.bstride = (long long)c->bstride; v.bfirst = 0; v.refine_below = refine_below();
    } else {
        // same padded geometry: clear the whole state, so no stale (possibly non-finite) value of a
        // previous problem can sit in a padding lane
        LP_HIP(hipMemsetAsync(c->arena, 0, c->arena_bytes, st));
    }
    c->m = m; c->n = n;
    c->ns = (int)n_slack; c->nx = (int)nx;
    c->va.n = (int)n; c->va.m = (int)m;
    c->va.n_total = (long long)n; c->va.gs = nullptr; c->colsplit = false;   // lpipm_upload_nsplit overrides
    // A single LP's loop ends on the host, so its kernels need not test the done word (one dependent load
    // less at the start of ~100 short kernels) -- except the head of an iteration, which is enqueued before
    // the host has seen the previous status.  In a batch every kernel tests it.
    c->bt = Batch{count, (long long)c->bstride, count > 1 ? c->va.done : nullptr};
    c->bt_head = Batch{count, (long long)c->bstride, c->va.done};
    c->va.done_chk = c->bt.done;
    std::vector<double> c0v((size_t)count, 0.0);          // must outlive the asynchronous copies below
    for (int i = 0; i < count; ++i) c0v[i] = c0 ? c0[i] : 0.0;
    if (parts) {   // rows of A_ub, then rows of A_eq; b likewise; c = [c; 0] (the arena is zero)
        const uint64_t m_ub = parts->m_ub, m_eq = m - m_ub;
        if (m_ub) {
            LP_HIP(hipMemcpy2DAsync(c->A, (size_t)npa * sizeof(double), parts->A_ub, (size_t)parts->lda_ub * sizeof(double),
                                    (size_t)nx * sizeof(double), (size_t)m_ub, hipMemcpyHostToDevice, st));
            LP_HIP(hipMemcpyAsync((void*)c->va.b, parts->b_ub, m_ub * sizeof(double), hipMemcpyHostToDevice, st));
        }
        if (m_eq) {
            LP_HIP(hipMemcpy2DAsync(c->A + (size_t)m_ub * npa, (size_t)npa * sizeof(double), parts->A_eq,
                                    (size_t)parts->lda_eq * sizeof(double), (size_t)nx * sizeof(double), (size_t)m_eq,
                                    hipMemcpyHostToDevice, st));
            LP_HIP(hipMemcpyAsync((void*)(c->va.b + m_ub), parts->b_eq, m_eq * sizeof(double), hipMemcpyHostToDevice, st));
        }
        LP_HIP(hipMemcpyAsync((void*)c->va.c, cc[0], nx * sizeof(double), hipMemcpyHostToDevice, st));
        LP_HIP(hipMemcpyAsync((void*)(c->va.S + S_C0), &c0v[0], sizeof(double), hipMemcpyHostToDevice, st));
    }
    for (int i = 0; i < count && !parts; ++i) {
        const size_t off = (size_t)i * c->bstride;
        LP_HIP(hipMemcpy2DAsync((char*)c->A + off, (size_t)npa * sizeof(double), A[i], (size_t)lda * sizeof(double),
                                (size_t)nx * sizeof(double), (size_t)m, hipMemcpyHostToDevice, st));
        LP_HIP(hipMemcpyAsync((char*)c->va.b + off, b[i], m * sizeof(double), hipMemcpyHostToDevice, st));
        LP_HIP(hipMemcpyAsync((char*)c->va.c + off, cc[i], n * sizeof(double), hipMemcpyHostToDevice, st));
        LP_HIP(hipMemcpyAsync((char*)(c->va.S + S_C0) + off, &c0v[i], sizeof(double), hipMemcpyHostToDevice, st));
    }
    LP_HIP(hipStreamSynchronize(st));   // the caller's arrays and c0v are free again from here
    c->has_problem = true;
    c->cnt_dirty = true;
    bind_status_pinned(c, true);
    return LPIPM_OK;
}

extern "C" int lpipm_upload(lpipm_ctx* c, uint64_t m, uint64_t n, const double* A, uint64_t lda,
                            const double* b, const double* cc, double c0) {
    return lpipm_upload_slack(c, m, n, A, lda, b, cc, c0, 0);
}

extern "C" int lpipm_upload_slack(lpipm_ctx* c, uint64_t m, uint64_t n, const double* A, uint64_t lda,
                                  const double* b, const double* cc, double c0, uint64_t n_slack) {
    return upload_impl(c, 1, m, n, &A, lda, &b, &cc, &c0, n_slack);
}

extern "C" int lpipm_upload_ub_eq(lpipm_ctx* c, uint64_t n, uint64_t m_ub, const double* A_ub, uint64_t lda_ub,
                                  const double* b_ub, uint64_t m_eq, const double* A_eq, uint64_t lda_eq,
                                  const double* b_eq, const double* cc, double c0) {
    if (m_ub + m_eq == 0) return LPIPM_UNCONSTRAINED;                       // linear_program.rs:134-136
    if (!cc || n == 0 || (m_ub && (!A_ub || !b_ub || lda_ub < n)) || (m_eq && (!A_eq || !b_eq || lda_eq < n)))
        return LPIPM_ERR_BAD_ARGUMENT;
    const UploadParts parts{m_ub, A_ub, lda_ub, b_ub, A_eq, lda_eq, b_eq};
    return upload_impl(c, 1, m_ub + m_eq, n + m_ub, nullptr, 0, nullptr, &cc, &c0, m_ub, &parts);
}

// ------------------------------------------------------------------------------------------------
// Y = add + A.W and Upart = row-split slabs of A^T.V on the stored structural columns, plus the
// identity block of the slack columns
static hipError_t ctx_gemv_n(lpipm_ctx* c, int nrhs, const double* W, const double* add0, const double* add1, double* Y,
                             const Batch& bt) {
    ++c->gemv_passes;
    hipError_t e = launch_gemv_n(c->A, c->npa, (int)c->m, c->npa, nrhs, W, c->np, add0, add1, Y, c->mp, c->st, 1.0, bt);
    if (e != hipSuccess) return e;
    return launch_slack_n(c->ns, c->nx, nrhs, W, c->np, Y, c->mp, c->st, bt);
}
static hipError_t ctx_gemv_t(lpipm_ctx* c, int nrhs, const double* V, const Batch& bt) {
    ++c->gemv_passes;
    hipError_t e = launch_gemv_t(c->A, c->npa, c->mp, c->npa, nrhs, V, c->mp, c->ATpart, c->st, c->np, bt);
    if (e != hipSuccess) return e;
    return launch_slack_t(c->ns, c->nx, nrhs, c->nsplit, V, c->mp, c->ATpart, c->np, c->st, bt);
}

// M = A . diag(dinv) . A^T, lower tiles (newton_equations.rs:54-57); a second copy of it goes to M0 (the matrix the
// refined Cholesky solves take their residuals against: M itself is factorised in place)
static GemmArgs adat_args(lpipm_ctx* c, const Batch& bt) {
    GemmArgs g{};
    g.P = c->A; g.ldp = c->npa; g.Q = c->A; g.ldq = c->npa; g.s = c->va.dinv;
    g.C = c->M; g.ldc = c->mp; g.K = c->npa; g.alpha = 1.0; g.beta = 0.0;
    g.ntiles = c->ntiles; g.tiles_lower = 1; g.ntj = 0; g.tile_list = c->tile_list;
    g.diag_pad_from = (int)c->m; g.ws = c->ws; g.nwg = c->adat_nwg; g.batch = bt; g.sk_claim = c->sk_claim; g.streamk = 1;
    g.C2 = (c->refine > 0 && gemm_streamk_split(c->npa / BK)) ? c->M0 : nullptr;    // only the refined solves need M itself
    return g;
}
static AdatUnitsArgs adat_units_args(lpipm_ctx* c, const Batch& bt) {
    AdatUnitsArgs a{};
    a.A = c->A; a.lda = c->npa; a.s = c->va.dinv; a.C = c->M; a.ldc = c->mp; a.K = c->npa;
    a.C2 = (c->refine > 0 && c->cpt > 1) ? c->M0 : nullptr;                // only the refined solves need M itself
    a.ntiles = c->ntiles; a.tile_list = c->tile_list; a.unit_list = c->unit_list; a.nunits = c->nunits; a.upc = c->upc;
    a.diag_pad_from = (int)c->m; a.slabs = c->ws; a.tile_cnt = c->tile_cnt;
    a.grp_cnt = nullptr; a.grp_w = POTRF_OUTER; a.batch = bt;
    return a;
}
// clears the arrival counters (tiles and groups) of every LP of the batch
static hipError_t clear_unit_counters(lpipm_ctx* c, const Batch& bt, hipStream_t st) {
    char* p = (char*)c->tile_cnt + (size_t)bt.first * (size_t)bt.stride;
    return bt.count == 1 ? hipMemsetAsync(p, 0, c->cnt_bytes, st)
                         : hipMemset2DAsync(p, (size_t)bt.stride, 0, c->cnt_bytes, (size_t)bt.count, st);
}
static hipError_t run_adat(lpipm_ctx* c, const Batch& bt) {
    hipError_t e;
    bool second_copy;
    if (c->units) {
        const AdatUnitsArgs a = adat_units_args(c, bt);
        if (c->cpt > 1 && c->cnt_dirty && (e = clear_unit_counters(c, bt, c->st)) != hipSuccess) return e;
        if ((e = launch_adat_units(a, c->st)) != hipSuccess) return e;
        c->cnt_dirty = false;            // the last arriver of every tile has put its word back to zero
        second_copy = a.C2 != nullptr;
    } else {
        const GemmArgs g = adat_args(c, bt);
        if ((e = launch_gemm_nt(g, c->st)) != hipSuccess) return e;
        second_copy = g.C2 != nullptr;
    }
    e = launch_slack_diag(c->ns, c->nx, c->va.dinv, c->M, c->mp, c->st, bt);   // + diag(D_slack)
    if (e != hipSuccess) return e;
    if (c->refine <= 0) return hipSuccess;
    if (second_copy) return launch_slack_diag(c->ns, c->nx, c->va.dinv, c->M0, c->mp, c->st, bt);
    vec_copy_lower(c->M, c->M0, c->mp, c->mp, c->st, bt);     // short contraction: one store per tile, copied afterwards
    return hipGetLastError();
}

// The normal equations AND their Cholesky factor, the factorisation running beside A.D.A^T (single LP, m >= 2048).
// Why: the factorisation is a chain of 32 (m = 4096) dependent steps -- one 128 x 128 diagonal block on ONE CU, its panel
// solve, the update of the next block -- with 1-30 of 256 CUs busy for 1.4 of its 1.9 ms.  It is given other work to hide
// behind, the 2.3 ms of A.D.A^T:
//   * A.D.A^T is ONE launch of (tile, chunk) units in COLUMN-GROUP-MAJOR order (group g = tile columns 4g .. 4g+3 = one
//     outer panel of the factorisation) on the throughput stream st_a (CU mask: all but R CUs per XCC).  The workgroup that
//     completes a tile stores it write-through and bumps its group's word; group g is complete when that word reaches the
//     group's tile count -- long before the launch ends (round 2 cut A.D.A^T into eight stream-K launches with eight tails
//     and eight fix-ups: 2.84 + 0.25 ms against 2.35, and lost);
//   * the factorisation is left-looking at the outer-panel level: before panel g is factorised, its column group gets the
//     updates of all panels before it in ONE product (K = 512 g, stream-K, alpha = -1, beta = 1) on st_u -- a second stream
//     with the throughput mask, so the update's workgroups take slots between A.D.A^T's units, which are still being
//     dispatched -- behind a one-wave kernel that waits for the group's word (launch_wait_count);
//   * inside a panel the chain is what it was (potrf_panel_chain), on the chain stream st_b (CU mask: the R reserved CUs
//     per XCC: a diagonal-block kernel needs a whole CU's LDS);
//   * events: panel g-1 done (st_b -> st_u), group g updated (st_u -> st_b).
// The results are those of the same factorisation run alone (fixed summation orders everywhere).
static int enqueue_factor_grouped(lpipm_ctx* c, const Batch& bt) {
    hipStream_t sm = c->st, sa = c->st_a, su = c->st_u, sb = c->st_b;
    const int ng = (int)c->grp_nt.size();
    const int wg_cus = c->num_cu - 8 * c->overlap_cus;
    // the words the other streams poll are cleared BEFORE they are released (a wait kernel that ran ahead of the memset
    // would see the previous iteration's full counts)
    LP_HIP(clear_unit_counters(c, bt, sm));
    c->cnt_dirty = true;
    LP_HIP(hipEventRecord(c->ev_fork, sm));
    LP_HIP(hipStreamWaitEvent(sa, c->ev_fork, 0));
    LP_HIP(hipStreamWaitEvent(su, c->ev_fork, 0));
    LP_HIP(hipStreamWaitEvent(sb, c->ev_fork, 0));
    LP_HIP(potrf_clear_info(c->va.potrf_info, sb, bt));
    const bool timed = c->profiling != 0;
    hipEvent_t* ev = c->ev_adat.data() + (c->overlap_sections & 1) * 2;   // {begin, end} of this section's launch
    {   // throughput stream: the whole of A.D.A^T, group by group
        AdatUnitsArgs a = adat_units_args(c, bt);
        a.tile_list = c->tile_list_grp; a.unit_list = c->unit_list_grp; a.nunits = c->nunits_grp;
        a.grp_cnt = c->grp_cnt;
        if (timed) LP_HIP(hipEventRecord(ev[0], sa));
        LP_HIP(launch_adat_units(a, sa));
        if (timed) LP_HIP(hipEventRecord(ev[1], sa));
        LP_HIP(hipEventRecord(c->ev_adat_done, sa));
    }
    for (int g = 0; g < ng; ++g) {
        const int J0 = g * POTRF_OUTER, J1 = J0 + POTRF_OUTER < c->mp / NB ? J0 + POTRF_OUTER : c->mp / NB;
        hipStream_t sg = g == 0 ? sb : su;           // group 0 needs no update: the chain stream waits for it itself
        if (g > 0) LP_HIP(hipStreamWaitEvent(su, c->ev_chain[g - 1], 0));
        LP_HIP(launch_wait_count(c->grp_cnt + g, (unsigned)c->grp_nt[g], bt.done, c->wait_timeout, sg));
        if (c->ns > J0 * NB) {                       // + diag(D_slack) on the group's diagonal blocks (before the update, as alone)
            const int r0 = J0 * NB, cnt = (c->ns < J1 * NB ? c->ns : J1 * NB) - r0;
            LP_HIP(launch_slack_diag(cnt, c->nx, c->va.dinv + r0, c->M + (size_t)r0 * (c->mp + 1), c->mp, sg, bt));
            if (c->refine > 0) LP_HIP(launch_slack_diag(cnt, c->nx, c->va.dinv + r0, c->M0 + (size_t)r0 * (c->mp + 1), c->mp, sg, bt));
        }
        if (g > 0) {   // ... minus what the panels before it contribute: C -= L[rows, 0:K) . L[cols, 0:K)^T
            GemmArgs u{};
            u.P = c->M; u.ldp = c->mp; u.Q = c->M; u.ldq = c->mp; u.s = nullptr;
            u.C = c->M; u.ldc = c->mp; u.K = J0 * NB; u.alpha = -1.0; u.beta = 1.0;
            u.ntiles = c->grp_nt[g]; u.tiles_lower = 1; u.tile_list = c->tile_list_grp + c->grp_off[g];
            u.diag_pad_from = -1; u.ws = c->ws_upd; u.batch = bt; u.sk_claim = c->sk_claim_upd; u.streamk = 1;
            u.nwg = gemm_streamk_nwg(u.ntiles, u.K / BK, wg_cus);
            LP_HIP(launch_gemm_nt(u, su));
            LP_HIP(hipEventRecord(c->ev_ready[g], su));
            LP_HIP(hipStreamWaitEvent(sb, c->ev_ready[g], 0));
        }
        LP_HIP(potrf_panel_chain(c->M, c->mp, c->mp, c->plan, c->va.potrf_info, sb, bt, J0, J1));
        LP_HIP(hipEventRecord(c->ev_chain[g], sb));
    }
    LP_HIP(hipStreamWaitEvent(sm, c->ev_chain[ng - 1], 0));
    LP_HIP(hipStreamWaitEvent(sm, c->ev_adat_done, 0));        // (complete by then: every group word was waited for)
    LP_HIP(potrf_superblock_inverses(c->plan, sm, bt));
    return LPIPM_OK;
}

// v = M^-1 r through the Cholesky factor (newton_equations.rs:151-169); optionally (LPIPM_REFINE, see lpipm_ctx::refine)
// with one step of iterative refinement against the matrix itself:  v0 = L^-T L^-1 r;  rho = r - M.v0 (doubled
// precision, one read of the lower triangle);  v = v0 + L^-T L^-1 rho.  R: nrhs x mp, in/out.
static int chol_solve_refined(lpipm_ctx* c, int nrhs, double* R, const Batch& bt) {
    hipStream_t st = c->st;
    if (!c->refine_now) { LP_HIP(launch_chol_solve(c->M, c->mp, c->plan, nrhs, R, c->Y, st, bt)); return LPIPM_OK; }
    // the refinement's launches skip an LP whose own word says so (a finished one, or one that does not need it yet)
    const Batch br = c->refine == 2 ? bt : Batch{bt.count, bt.stride, c->va.skip_refine, bt.first};
    vec_rows_copy(c->mp, nrhs, c->R0, R, st, br);
    LP_HIP(launch_chol_solve(c->M, c->mp, c->plan, nrhs, R, c->Y, st, bt));
    LP_HIP(launch_symv_residual(c->M0, c->mp, c->mp, nrhs, R, c->mp, c->R0, c->mp, c->Rho, c->mp, c->symv_ws, st, br));
    LP_HIP(launch_chol_solve(c->M, c->mp, c->plan, nrhs, c->Rho, c->Y, st, br));
    vec_rows_add(c->mp, nrhs, R, c->Rho, st, br);
    LP_HIP(hipGetLastError());
    return LPIPM_OK;
}

static int enqueue_residuals(lpipm_ctx* c, int is_init, int ip_next, double tol) {
    VecArgs& v = c->va;
    // A.x and A^T.y at the current point (residual.rs:23,25)
    XRank xr{xrank_fn, c};
    if (!(c->colsplit && c->world > 1)) {       // both products in one read of A
        ++c->gemv_passes;
        v.ax_chunks = gemv_dual_chunks(c->npa);
        LP_HIP(launch_gemv_dual(c->A, c->npa, c->mp, c->npa, v.x, v.y, v.Ax, c->ATpart, c->np, c->st, c->bt));
        LP_HIP(launch_slack_n(c->ns, c->nx, 1, v.x, c->np, v.Ax, c->mp, c->st, c->bt));          // into chunk slab 0
        LP_HIP(launch_slack_t(c->ns, c->nx, 1, c->nsplit, v.y, c->mp, c->ATpart, c->np, c->st, c->bt));
    } else {
        v.ax_chunks = 1;
        LP_HIP(ctx_gemv_n(c, 1, v.x, nullptr, nullptr, v.Ax, c->bt));
        LP_TRY(ctx_allreduce(c, v.Ax, c->m, 0));          // n-split: A.x = sum over ranks of A_g.x_g
        LP_HIP(ctx_gemv_t(c, 1, v.y, c->bt));
    }
    prof_mark(c, T_GEMV);
    // small LPs: the launch goes on with the next iteration's Dinv / r_hat set-up (enqueue_head then skips it); not under graph
    // replay, where the head must be the same launches every time
    v.status_seq = (int)(++c->seq_counter & 0x7fffffffu);
    const bool with_pred = !c->colsplit && vec_fused(v) && c->use_graph != 1;
    LP_TRY(vec_residuals(v, is_init, ip_next, tol, c->st, c->colsplit ? &xr : nullptr, with_pred));
    c->pred_done = with_pred;
    LP_HIP(hipGetLastError());
    return LPIPM_OK;
}

// status records of all LPs of the context -> pinned host array (96 bytes each)
// The status records go straight from the kernels that write them into the context's coherent pinned array
// (VecArgs::status_pinned: payload, system fence, sequence word), and a solve that needs nothing else from the stream at
// that point waits for an iteration by watching the sequence words (wait_status): neither a D2H copy launch nor an event
// record sits between the indicators and the next iteration's A.D.A^T (C2: 4 + 6 us of ~310 per iteration).
// LPIPM_STATUS_COPY=1: copy launch + event as in rounds 1-2.
static void bind_status_pinned(lpipm_ctx* c, bool allow) {
    void* dp = nullptr;
    const char* e = lp_knob("LPIPM_STATUS_COPY");
    const bool on = allow && !(e && e[0] == '1') && c->status_host && hipHostGetDevicePointer(&dp, c->status_host, 0) == hipSuccess;
    (void)hipGetLastError();
    c->va.status_pinned = on ? (StatusRec*)dp : nullptr;
}
// Waits until the records of the LPs `idx[0 .. count)` (nullptr: record 0) carry the sequence number of the last residual
// launch.  Spins on the pinned records (bounded: ~10 s, then the stream is drained and the records are checked once more).
static int wait_status(lpipm_ctx* c, const int* idx, int count) {
    if (!c->spin_status) { LP_HIP(hipEventSynchronize(c->ev_status)); return LPIPM_OK; }
    const int32_t want = (int32_t)c->va.status_seq;
    auto arrived = [&]() {
        for (int k = 0; k < count; ++k) {
            const StatusRec* r = c->status_host + (idx ? idx[k] : 0);
            if (__atomic_load_n(&r->pad_, __ATOMIC_ACQUIRE) != want) return false;
        }
        return true;
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (arrived()) return LPIPM_OK;
        __builtin_ia32_pause();
        if ((spins & 0xffffu) == 0xffffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) break;
    }
    LP_HIP(hipStreamSynchronize(c->st));
    if (arrived()) return LPIPM_OK;
    g_err_detail = "the status record of an iteration never reached the host";
    return LPIPM_ERR_HIP;
}
static int copy_status(lpipm_ctx* c) {
    if (c->va.status_pinned) {}          // written by the kernels themselves
    else if (c->B == 1) LP_HIP(hipMemcpyAsync(c->status_host, (const char*)c->va.status + (size_t)c->bt.first * c->bstride, sizeof(StatusRec), hipMemcpyDeviceToHost, c->st));
    else LP_HIP(hipMemcpy2DAsync(c->status_host, sizeof(StatusRec), (const char*)c->va.status + (size_t)c->bt.first * c->bstride, c->bstride,
                                 sizeof(StatusRec), (size_t)c->B, hipMemcpyDeviceToHost, c->st));
    if (c->factor_in_head || (c->colsplit && c->grouped_reduce))   // a wait kernel that gave up (its producer never ran) says so here
        LP_HIP(hipMemcpyAsync(c->timeout_host, c->wait_timeout, sizeof(unsigned int), hipMemcpyDeviceToHost, c->st));
    return LPIPM_OK;
}

// one IPM iteration: get_delta (feasible_point.rs:110-152), step length (mod.rs:216-221),
// do_step (:222), indicators (:225)
// Head of an iteration: Dinv = x/z and the normal equations M = A.Dinv.A^T (newton_equations.rs:54-57).  It needs
// nothing from the host, and its kernels test the LP's done word, so it may be enqueued BEFORE the host has read
// the status of the previous iteration: the GPU goes straight from one iteration into the big kernel of the
// next instead of idling through the read-back, and if the LP turns out to be finished the two kernels return
// at once.
static int enqueue_head(lpipm_ctx* c) {
    hipStream_t st = c->st;
    VecArgs vh = c->va;
    vh.done_chk = c->bt_head.done;
    prof_mark(c, T_VEC);
    if (c->pred_done) c->pred_done = false;       // the residual launch in front of this head has done it (enqueue_residuals)
    else vec_pred_setup(vh, st);
    if (c->factor_in_head) {   // A.D.A^T and the Cholesky factorisation side by side (newton_equations.rs:55-57, :129-131)
        prof_mark(c, T_VEC);
        LP_TRY(enqueue_factor_grouped(c, c->bt_head));
        prof_mark(c, T_POTRF);
        c->overlap_sections++;
        return LPIPM_OK;
    }
    prof_mark(c, T_VEC, true);
    if (c->colsplit && c->world > 1 && c->grouped_reduce && c->st_c) {
        // n-split, M = sum_g A_g D_g A_g^T, PIPELINED: one A.D.A^T launch in column-group-major order on the solver's stream;
        // the workgroup that completes a group's last tile bumps the group's word; on the communication stream a one-wave
        // kernel waits for that word, the group's tiles are packed, summed over the ranks (the caller's all-reduce) and
        // unpacked -- while the launch goes on with the next groups.  After the last tile only the last group's sum is
        // left (C5: 1/32 .. 1/8 of the 1.08 GB that round 2 reduced in one block after the launch).  Element-wise sums:
        // the same values as one reduction of the whole triangle.
        hipStream_t sc = c->st_c;
        const Batch& bt = c->bt_head;
        LP_HIP(clear_unit_counters(c, bt, st));
        c->cnt_dirty = true;
        LP_HIP(hipEventRecord(c->ev_c0, st));
        LP_HIP(hipStreamWaitEvent(sc, c->ev_c0, 0));
        AdatUnitsArgs a = adat_units_args(c, bt);
        a.tile_list = c->tile_list_grp; a.unit_list = c->unit_list_grp; a.nunits = c->nunits_grp; a.upc = 1;
        a.grp_cnt = c->grp_cnt; a.C2 = nullptr;
        LP_HIP(launch_adat_units(a, st));
        for (size_t g = 0; g < c->grp_nt.size(); ++g) {
            double* slice = c->mpack + (size_t)c->grp_off[g] * TILE * TILE;
            LP_HIP(launch_wait_count(c->grp_cnt + g, (unsigned)c->grp_nt[g], bt.done, c->wait_timeout, sc));
            vec_pack_tiles(c->M, c->mp, c->tile_list_grp + c->grp_off[g], c->grp_nt[g], slice, 0, sc);
            LP_TRY(ctx_allreduce(c, slice, (uint64_t)c->grp_nt[g] * TILE * TILE, 0, sc));
            vec_pack_tiles(c->M, c->mp, c->tile_list_grp + c->grp_off[g], c->grp_nt[g], slice, 1, sc);
        }
        LP_HIP(hipEventRecord(c->ev_c1, sc));
        LP_HIP(hipStreamWaitEvent(st, c->ev_c1, 0));
        if (c->refine > 0) vec_copy_lower(c->M, c->M0, c->mp, c->mp, st, c->bt_head);   // the summed matrix, for the refined solves
        prof_mark(c, T_ADAT, true);
        return LPIPM_OK;
    }
    LP_HIP(run_adat(c, c->bt_head));                                       // newton_equations.rs:55-57
    if (c->colsplit && c->world > 1) {                                     // n-split: M = sum_g A_g D_g A_g^T
        vec_pack_lower(c->M, c->mp, c->mp, c->mpack, 0, st);
        LP_TRY(ctx_allreduce(c, c->mpack, c->mpack_count, 0));
        vec_pack_lower(c->M, c->mp, c->mp, c->mpack, 1, st);
        if (c->refine > 0) vec_copy_lower(c->M, c->M0, c->mp, c->mp, st, c->bt_head);   // the summed matrix, for the refined solves
    }
    prof_mark(c, T_ADAT, true);
    return LPIPM_OK;
}

// The look-ahead of launch_potrf needs a second stream: not while a graph is being captured on the solver's stream.
static const PotrfLookahead* lookahead(lpipm_ctx* c) { return (c->la.side && c->use_graph != 1) ? &c->la : nullptr; }

// The rest of the iteration, ending with the status record on its way to the host and ev_status behind it.
static int enqueue_tail(lpipm_ctx* c, int ip, const lpipm_opts* o) {
    VecArgs& v = c->va;
    hipStream_t st = c->st;
    XRank xr_{xrank_fn, c};
    const XRank* xr = c->colsplit ? &xr_ : nullptr;
    const Batch& bt = c->bt;
    const bool chol = o->solver_type == LPIPM_SOLVER_CHOLESKY;
    if (c->factor_in_head) {}                                                             // factorised beside A.D.A^T
    // (no clearing of the pivot-failure word: k_blind_start and every k_scalar_indicators leave it zero)
    else if (chol) LP_HIP(launch_potrf(c->M, c->mp, c->mp, c->plan, v.potrf_info, st, bt, lookahead(c), false));   // :129-131
    else           LP_HIP(launch_qr_factor(c->M, c->mp, c->mp, c->tau, v.potrf_info, st));   // :133-149
    prof_mark(c, T_POTRF);
    // predictor: both sym_solve calls of solve_newton_equations (:187-188) in one pass each
    if (!c->colsplit) {
        LP_HIP(ctx_gemv_n(c, 2, v.W, v.b, v.rP, v.R, bt));  // :220
    } else {   // the addend r2 enters once, after the cross-rank sum of the column-split products
        LP_HIP(ctx_gemv_n(c, 2, v.W, nullptr, nullptr, v.R, bt));
        LP_TRY(ctx_allreduce(c, v.R, (uint64_t)2 * c->mp, 0));
        vec_add_rows((int)c->m, 2, v.R, c->mp, v.b, v.rP, st);
    }
    prof_mark(c, T_GEMV);
    if (chol) LP_TRY(chol_solve_refined(c, 2, v.R, bt));                                    // :221, :154
    else      LP_HIP(launch_qr_solve(c->M, c->mp, c->mp, c->tau, 2, v.R, v.potrf_info, st));   // :155-166
    prof_mark(c, T_TRSV);
    LP_HIP(ctx_gemv_t(c, 2, v.R, bt));              // :223
    prof_mark(c, T_GEMV);
    if (!xr && vec_fused(v)) vec_fused_predictor(v, ip, st);   // the three below in one launch (small n)
    else {
    LP_TRY(vec_pq_uv(v, st, xr));                       // :223, delta.rs:29-32,38
    LP_TRY(vec_delta(v, 0, ip, 1.0, st, xr));           // delta.rs:33-37, feasible_point.rs:134-136
    vec_corr_setup(v, ip, st);              // rhat.rs:37-75
    }
    prof_mark(c, T_VEC);
    // corrector: only the second sym_solve changes
    if (!c->colsplit) {
        LP_HIP(ctx_gemv_n(c, 1, v.W, v.rP2, nullptr, v.R, bt));
    } else {
        LP_HIP(ctx_gemv_n(c, 1, v.W, nullptr, nullptr, v.R, bt));
        LP_TRY(ctx_allreduce(c, v.R, c->mp, 0));
        vec_add_rows((int)c->m, 1, v.R, c->mp, v.rP2, nullptr, st);
    }
    prof_mark(c, T_GEMV);
    if (chol) LP_TRY(chol_solve_refined(c, 1, v.R, bt));
    else      LP_HIP(launch_qr_solve(c->M, c->mp, c->mp, c->tau, 1, v.R, v.potrf_info, st));
    prof_mark(c, T_TRSV);
    LP_HIP(ctx_gemv_t(c, 1, v.R, bt));
    prof_mark(c, T_GEMV);
    if (!xr && vec_fused(v)) vec_fused_corrector(v, ip, o->alpha0, st);
    else {
    LP_TRY(vec_uv_corr(v, st, xr));
    LP_TRY(vec_delta(v, 1, ip, o->alpha0, st, xr));     // mod.rs:216-221
    vec_step(v, ip, o->alpha0, st);         // feasible_point.rs:76-106 (+ the step length, mod.rs:216-221, when folded)
    }
    prof_mark(c, T_VEC);
    LP_TRY(enqueue_residuals(c, 0, 0, o->tol));   // mod.rs:225
    LP_TRY(copy_status(c));
    prof_mark(c, T_VEC);
    if (!c->spin_status) LP_HIP(hipEventRecord(c->ev_status, st));
    return LPIPM_OK;
}

static int enqueue_iteration(lpipm_ctx* c, int ip, const lpipm_opts* o) {
    LP_TRY(enqueue_head(c));
    return enqueue_tail(c, ip, o);
}

// The ~100 launches of one iteration replayed as one hipGraph launch: the sequence and every argument
// are the same from iteration to iteration (only `ip` differs, on the first one).  Capturing does not
// execute anything, so the first use of a key costs one capture + instantiate and then runs the graph.
static int run_iteration(lpipm_ctx* c, int ip, const lpipm_opts* o) {
    const bool graphable = c->use_graph == 1 && !c->profiling && !c->colsplit && o->solver_type == LPIPM_SOLVER_CHOLESKY;
    if (!graphable) return enqueue_iteration(c, ip, o);
    for (auto& g : c->graphs)
        if (g.ip == ip && g.refine == (int)c->refine_now && g.alpha0 == o->alpha0 && g.tol == o->tol) {
            LP_HIP(hipGraphLaunch(g.exec, c->st));
            return LPIPM_OK;
        }
    LP_HIP(hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_iteration(c, ip, o);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c->st, &graph);
    if (rc != LPIPM_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    LP_HIP(e);
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    LP_HIP(ei);
    c->graphs.push_back({ip, (int)c->refine_now, o->alpha0, o->tol, exec});
    LP_HIP(hipGraphLaunch(exec, c->st));
    return LPIPM_OK;
}

static void print_row(double alpha, const StatusRec& s) {  // mod.rs:228 + indicators.rs:25-33
    printf("%.8f\t%.8f\t%.8f\t%.8f\t%.8f\t%8.3f\n", alpha, s.rho_p, s.rho_d, s.rho_g, s.rho_mu, s.obj);
}

static int solve_impl(lpipm_ctx* c, const lpipm_opts* o, double* x_host, void* x_dev, double* fun_out,
                      uint64_t* iters_out, lpipm_iter_row* log) {
    if (!c || !o) return LPIPM_ERR_BAD_ARGUMENT;
    // InteriorPointBuilder::build, mod.rs:118-128
    if (!(o->alpha0 > 0.0) || !(o->alpha0 < 1.0)) return LPIPM_INVALID_PARAMETER;
    if (!(o->tol > 0.0)) return LPIPM_INVALID_PARAMETER;
    if (o->solver_type < 0 || o->solver_type > 2) return LPIPM_INVALID_PARAMETER;
    if (!c->has_problem) return LPIPM_ERR_NO_PROBLEM;
    if (c->B != 1) return LPIPM_ERR_BAD_ARGUMENT;   // a lockstep batch is solved by solve_lockstep
    if (o->solver_type != LPIPM_SOLVER_CHOLESKY && c->mp > 16384) return LPIPM_ERR_UNSUPPORTED;  // QR solve keeps the rhs in LDS
    LP_HIP(hipSetDevice(c->device));
    VecArgs& v = c->va;
    hipStream_t st = c->st;
    if (c->use_graph < 0) {   // measurement knobs: LPIPM_GRAPH=1 (hipGraph replay), LPIPM_SPECULATE=0 (no early head)
        const char* g = lp_knob("LPIPM_GRAPH");
        c->use_graph = (g && g[0] == '1') ? 1 : 0;
        const char* sp = lp_knob("LPIPM_SPECULATE");
        c->no_speculate = sp && sp[0] == '0';
    }
    c->factor_in_head = c->overlap && !c->colsplit && c->use_graph != 1 && o->solver_type == LPIPM_SOLVER_CHOLESKY;
    c->overlap_sections = 0;
    for (int t = 0; t < T_NTAGS; ++t) c->tag_ms[t] = 0.0;
    c->times = lpipm_phase_times{};
    c->nmarks = 0;
    c->gemv_passes = 0;
    uint64_t adat_launches = 0;
    if (c->profiling) LP_HIP(hipEventRecord(c->ev_begin, st));

    vec_blind_start(v, st);                               // feasible_point.rs:24-31
    prof_mark(c, T_VEC);
    LP_TRY(enqueue_residuals(c, 1, o->ip ? 1 : 0, o->tol));  // feasible_point.rs:32, mod.rs:206
    LP_TRY(copy_status(c));
    prof_mark(c, T_VEC);
    // the host needs the starting point's indicators only for the `disp` table, for the selective refinement's first decision
    // and for the phase marks: otherwise the first iteration is enqueued without a round trip to the host (~25 us per solve)
    const bool need_start_row = o->disp || c->refine == 1 || c->profiling == 1;   // (profiling 2 brackets A.D.A^T only: no mark yet)
    if (need_start_row) {
        LP_HIP(hipStreamSynchronize(st));
        prof_collect(c);
    }
    if (o->disp) {                                        // mod.rs:208-211
        printf("alpha     \trho_p     \trho_d     \trho_g     \trho_mu    \tobj       \n");
        print_row(1.0, *c->status_host);
    }
    c->refine_now = c->refine == 2 || (c->refine == 1 && c->status_host->rho_mu <= refine_below());
    int ip = o->ip ? 1 : 0;
    int ret = LPIPM_ITERATION_LIMIT;
    uint64_t iteration = 0;
    // the head of iteration k+1 goes out before the status of iteration k is read (see enqueue_head); not when the
    // iteration is replayed as a graph or contains host-side collectives
    const bool speculate = c->use_graph != 1 && !c->colsplit && !c->no_speculate;
    // (with every phase bracketed -- profiling 1 -- the last mark of an iteration is recorded BEHIND the indicators kernel and
    //  has to have completed when it is read: the event wait stays; profiling 2's two marks sit in front of it)
    c->spin_status = speculate && c->va.status_pinned != nullptr && c->profiling != 1;
    bool head_out = false;
    for (iteration = 1; iteration <= o->max_iter; ++iteration) {   // mod.rs:213
        if (!speculate) {
            LP_TRY(run_iteration(c, ip, o));
            LP_HIP(hipStreamSynchronize(st));
            prof_collect(c);
            prof_collect_overlap(c, c->overlap_sections - 1);
        } else {
            if (!head_out) LP_TRY(enqueue_head(c));
            LP_TRY(enqueue_tail(c, ip, o));
            const size_t marks = c->nmarks;
            head_out = iteration < o->max_iter;
            if (head_out) LP_TRY(enqueue_head(c));
            LP_TRY(wait_status(c, nullptr, 1));
            prof_collect(c, marks);
            prof_collect_overlap(c, c->overlap_sections - (head_out ? 2 : 1));
        }
        ++adat_launches;
        if ((c->factor_in_head || (c->colsplit && c->grouped_reduce)) && *c->timeout_host != 0) {
            g_err_detail = "a column group of A.D.A^T did not complete within the wait kernel's bound";
            LP_HIP(hipStreamSynchronize(st));
            return LPIPM_ERR_HIP;
        }
        const StatusRec s = *c->status_host;
        // EquationSolverType::build failure (newton_equations.rs:58-63) and the NaN check on p, q
        // (:190-194) both surface as NumericalProblem from get_delta (mod.rs:215)
        if (s.potrf_info != 0 || (s.flags & FLAG_NAN_PQ)) { ret = LPIPM_NUMERICAL_PROBLEM; break; }
        c->refine_now = c->refine == 2 || (c->refine == 1 && s.rho_mu <= refine_below());   // the next iteration's solves
        ip = 0;                                                    // mod.rs:223
        if (o->disp) print_row(s.alpha, s);
        if (log) {
            lpipm_iter_row& r = log[iteration - 1];
            r.alpha = s.alpha; r.rho_p = s.rho_p; r.rho_d = s.rho_d; r.rho_A = s.rho_A;
            r.rho_g = s.rho_g; r.rho_mu = s.rho_mu; r.obj = s.obj;
        }
        if (s.status == ST_OPTIMAL) { ret = LPIPM_OK; break; }             // mod.rs:231
        if (s.status == ST_INFEASIBLE) { ret = LPIPM_INFEASIBLE; break; }   // :232
        if (s.status == ST_UNBOUNDED) { ret = LPIPM_UNBOUNDED; break; }     // :233
    }
    if (ret == LPIPM_ITERATION_LIMIT) iteration = o->max_iter;
    if (ret == LPIPM_OK || ret == LPIPM_ITERATION_LIMIT) {
        XRank xrf{xrank_fn, c};
        LP_TRY(vec_final_x(v, c->xout, st, c->colsplit ? &xrf : nullptr));   // mod.rs:231/238, :165
        LP_HIP(hipGetLastError());
        if (x_dev) LP_HIP(hipMemcpyAsync(x_dev, c->xout, c->n * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (x_host) {   // through a pinned buffer: truly asynchronous, one synchronisation for x and the status record
            if (c->x_pinned_cap < c->n) {
                if (c->x_pinned) (void)hipHostFree(c->x_pinned);
                c->x_pinned = nullptr; c->x_pinned_cap = 0;
                LP_HIP(hipHostMalloc((void**)&c->x_pinned, c->n * sizeof(double)));
                c->x_pinned_cap = c->n;
            }
            LP_HIP(hipMemcpyAsync(c->x_pinned, c->xout, c->n * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        LP_TRY(copy_status(c));
        if (c->profiling) LP_HIP(hipEventRecord(c->ev_end, st));
        LP_HIP(hipStreamSynchronize(st));
        if (x_host) std::memcpy(x_host, c->x_pinned, c->n * sizeof(double));
        if (fun_out) *fun_out = c->status_host->obj;
    } else {
        if (c->profiling) LP_HIP(hipEventRecord(c->ev_end, st));
        LP_HIP(hipStreamSynchronize(st));
    }
    if (iters_out) *iters_out = iteration;
    if (c->profiling) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->ev_begin, c->ev_end);
        c->times.total_ms = ms;
        c->times.adat_ms = c->tag_ms[T_ADAT]; c->times.potrf_ms = c->tag_ms[T_POTRF];
        if (c->factor_in_head && c->profiling == 1) {   // T_POTRF spans the whole side-by-side section: what A.D.A^T does not hide
            c->times.potrf_ms = c->tag_ms[T_POTRF] - c->tag_ms[T_ADAT];
            if (c->times.potrf_ms < 0.0) c->times.potrf_ms = 0.0;
        }
        c->times.trsv_ms = c->tag_ms[T_TRSV]; c->times.gemv_ms = c->tag_ms[T_GEMV];
        c->times.vec_ms = c->tag_ms[T_VEC];
        c->times.adat_launches = adat_launches; c->times.iterations = iteration;
        // the speculatively enqueued head of the iteration after the last holds no GEMV pass: every counted pass ran
        c->times.gemv_passes = c->gemv_passes;
    }
    return ret;
}

extern "C" int lpipm_solve(lpipm_ctx* c, const lpipm_opts* o, double* x_slack_out, double* fun_out,
                           uint64_t* iterations_out, lpipm_iter_row* log) {
    if (!x_slack_out) return LPIPM_ERR_BAD_ARGUMENT;
    return solve_impl(c, o, x_slack_out, nullptr, fun_out, iterations_out, log);
}
extern "C" int lpipm_solve_device(lpipm_ctx* c, const lpipm_opts* o, void* x_dev_out, double* fun_out,
                                  uint64_t* iterations_out, lpipm_iter_row* log) {
    return solve_impl(c, o, nullptr, x_dev_out, fun_out, iterations_out, log);
}

// ------------------------------------------------------------------------------------------------
// ---- half-batch views -------------------------------------------------------------------------------------------------
static void destroy_views(lpipm_ctx* c) {
    for (lpipm_ctx* v : c->halves) {
        (void)hipSetDevice(v->device);
        if (v->st) { (void)hipStreamSynchronize(v->st); (void)hipStreamDestroy(v->st); }
        for (hipEvent_t e : v->events) (void)hipEventDestroy(e);
        if (v->ev_begin) (void)hipEventDestroy(v->ev_begin);
        if (v->ev_end) (void)hipEventDestroy(v->ev_end);
        if (v->ev_status) (void)hipEventDestroy(v->ev_status);
        if (v->status_host) (void)hipHostFree(v->status_host);
        if (v->timeout_host) (void)hipHostFree(v->timeout_host);
        v->plan = FactorPlan{};            // shares the parent's descriptors: never destroyed here
        delete v;
    }
    c->halves.clear();
}
// A view of the LPs [first, first + count) of c's resident batch: the same device state (every pointer stays LP 0's), its own
// stream, events and pinned status records.
static lpipm_ctx* make_view(const lpipm_ctx* c, int first, int count) {
    lpipm_ctx* v = new lpipm_ctx(*c);
    v->is_view = true;
    v->cnt_dirty = true;
    v->halves.clear(); v->workers.clear(); v->graphs.clear(); v->kallocs.clear();
    v->events.clear(); v->mark_tags.clear(); v->nmarks = 0;
    v->ev_ready.clear(); v->ev_chain.clear(); v->ev_adat.clear(); v->la = PotrfLookahead{};
    v->st = nullptr; v->st_a = v->st_b = v->st_u = nullptr; v->ev_fork = v->ev_adat_done = nullptr; v->overlap = false;
    v->ev_begin = v->ev_end = v->ev_status = nullptr; v->status_host = nullptr; v->timeout_host = nullptr;
    v->x_pinned = nullptr; v->x_pinned_cap = 0;
    v->mpack = nullptr; v->kM = v->kM0 = v->kR = v->kY = nullptr; v->kmp = 0; v->kplan = FactorPlan{};
    v->profiling = 0;
    v->B = count;
    v->bt = Batch{count, (long long)c->bstride, c->va.done, first};
    v->bt_head = v->bt;
    v->va.bcount = count; v->va.bfirst = first; v->va.done_chk = c->va.done;
    v->status_cap = (size_t)count;
    if (hipStreamCreateWithFlags(&v->st, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&v->status_host, (size_t)count * sizeof(StatusRec), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void**)&v->timeout_host, sizeof(unsigned int)) != hipSuccess ||
        hipEventCreate(&v->ev_begin) != hipSuccess || hipEventCreate(&v->ev_end) != hipSuccess ||
        hipEventCreateWithFlags(&v->ev_status, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        lpipm_ctx tmp_owner;               // release what was made
        tmp_owner.halves.push_back(v);
        destroy_views(&tmp_owner);
        return nullptr;
    }
    *v->timeout_host = 0;
    std::memset(v->status_host, 0, (size_t)count * sizeof(StatusRec));
    v->seq_counter = 0; v->spin_status = false;
    bind_status_pinned(v, c->va.status_pinned != nullptr);
    return v;
}

// Lockstep batch: B LPs of one shape resident at once (upload_impl with count = B), every launch of the
// iteration covering all of them (gridDim.z = B).  The ~100 dependent launches per iteration -- the
// latency floor of a small LP -- are then paid once per B LPs.  LPs finish at different iterations:
// k_scalar_indicators sets an LP's `done` word on the conditions that end the reference's loop
// (mod.rs:215, :231-233) and every later kernel skips it, so its iterate stays what it was; the host
// mirrors the same decisions from the status records to count iterations and pick the return codes.
// Where the solutions of a batch go: per-member host pointers, or rows of one device buffer.
struct XOut {
    double* const* host = nullptr;
    char* dev = nullptr;
    size_t stride_bytes = 0;
    bool valid() const { return host || dev; }
};
static int solve_lockstep_one(lpipm_ctx* c, const lpipm_opts* o, const XOut& xo, const uint64_t* rows, double* fun_out,
                              uint64_t* its_out, int32_t* status_out);
// A batch of at least 16 members is solved as TWO half-batches, each by its own host thread on its own stream (views of the
// context): the halves drift out of phase, and one half's A.D.A^T (MFMA-bound, fills the chip) runs beside the other half's
// factorisation chain, solves and passes over A (latency- and HBM-bound).  Every member goes through exactly the kernels
// and arguments of the one-stream path: the results are bit-identical (tests/test_gpu_c4_members.py).  Measured on the C4
// shard (32 x 1024x2048): +3 .. +6.5 % (profiles/r03_rejected_experiments.txt has the variants).  Not while profiling (the
// phase marks are per stream) and not for the views themselves.
static int solve_lockstep(lpipm_ctx* c, const lpipm_opts* o, const XOut& xo, const uint64_t* rows, double* fun_out,
                          uint64_t* its_out, int32_t* status_out) {
    if (!c || !o || !xo.valid() || !status_out) return LPIPM_ERR_BAD_ARGUMENT;
    if (c->is_view || !c->halves_env || c->profiling || c->B < 16 || !c->has_problem || c->colsplit)
        return solve_lockstep_one(c, o, xo, rows, fun_out, its_out, status_out);
    LP_HIP(hipSetDevice(c->device));
    if (c->halves.empty()) {
        const int h = c->B / 2;
        lpipm_ctx* a = make_view(c, 0, h);
        lpipm_ctx* b = a ? make_view(c, h, c->B - h) : nullptr;
        if (!a || !b) {
            if (a) { c->halves.push_back(a); destroy_views(c); }
            return solve_lockstep_one(c, o, xo, rows, fun_out, its_out, status_out);
        }
        c->halves.push_back(a); c->halves.push_back(b);
    }
    LP_HIP(hipStreamSynchronize(c->st));         // the upload (or whatever else the caller enqueued) precedes both halves
    int rc[2] = {LPIPM_OK, LPIPM_OK};
    // a view numbers its members from 0: member i of half k is member first + i of the batch
    std::vector<uint64_t> ident;
    if (!rows) { ident.resize((size_t)c->B); for (int i = 0; i < c->B; ++i) ident[(size_t)i] = (uint64_t)i; rows = ident.data(); }
    auto run = [&](int k) {
        lpipm_ctx* v = c->halves[(size_t)k];
        const int f = v->bt.first;
        rc[k] = solve_lockstep_one(v, o, xo, rows + f, fun_out ? fun_out + f : nullptr, its_out ? its_out + f : nullptr, status_out + f);
    };
    std::thread other(run, 1);
    run(0);
    other.join();
    return rc[0] != LPIPM_OK ? rc[0] : rc[1];
}

static int solve_lockstep_one(lpipm_ctx* c, const lpipm_opts* o, const XOut& xo, const uint64_t* rows, double* fun_out,
                              uint64_t* its_out, int32_t* status_out) {
    if (!c || !o || !xo.valid() || !status_out) return LPIPM_ERR_BAD_ARGUMENT;
    if (!(o->alpha0 > 0.0) || !(o->alpha0 < 1.0)) return LPIPM_INVALID_PARAMETER;   // mod.rs:118-128
    if (!(o->tol > 0.0)) return LPIPM_INVALID_PARAMETER;
    if (o->solver_type != LPIPM_SOLVER_CHOLESKY) return LPIPM_ERR_UNSUPPORTED;      // the QR arms are single-LP
    if (!c->has_problem) return LPIPM_ERR_NO_PROBLEM;
    if (c->colsplit) return LPIPM_ERR_UNSUPPORTED;
    LP_HIP(hipSetDevice(c->device));
    const int B = c->B;
    VecArgs& v = c->va;
    hipStream_t st = c->st;
    c->factor_in_head = false;
    // profiling (lpipm_set_profiling): the same event marks as a single solve; a phase's time is that of the whole batch's launch
    for (int t = 0; t < T_NTAGS; ++t) c->tag_ms[t] = 0.0;
    c->times = lpipm_phase_times{};
    c->nmarks = 0;
    c->gemv_passes = 0;
    uint64_t batch_iterations = 0;
    if (c->profiling) LP_HIP(hipEventRecord(c->ev_begin, st));
    vec_blind_start(v, st);                                                  // feasible_point.rs:24-31
    prof_mark(c, T_VEC);
    LP_TRY(enqueue_residuals(c, 1, o->ip ? 1 : 0, o->tol));                  // feasible_point.rs:32, mod.rs:206
    prof_mark(c, T_VEC);
    if (c->profiling) {                  // (nothing of the starting point is read by the host otherwise)
        LP_HIP(hipStreamSynchronize(st));
        prof_collect(c);
    }
    std::vector<int> ret((size_t)B, -1);                                     // -1: still iterating
    std::vector<uint64_t> its((size_t)B, 0);
    int running = B, ip = o->ip ? 1 : 0;
    bool head_out = false;
    c->spin_status = c->va.status_pinned != nullptr && !c->profiling;
    c->refine_now = c->refine == 2;      // (selective mode) at the starting point mu / mu_0 = 1: no member refines its first iteration
    for (uint64_t iteration = 1; iteration <= o->max_iter && running > 0; ++iteration) {   // mod.rs:213
        if (!head_out) LP_TRY(enqueue_head(c));
        LP_TRY(enqueue_tail(c, ip, o));
        const size_t marks = c->nmarks;
        head_out = iteration < o->max_iter;
        if (head_out) LP_TRY(enqueue_head(c));       // next iteration's A.D.A^T, before this one's status is read
        {   // the members that were still running when this iteration was enqueued write a record; the others are skipped
            std::vector<int> act;
            for (int i = 0; i < B; ++i) if (ret[i] < 0) act.push_back(i);
            LP_TRY(wait_status(c, act.data(), (int)act.size()));
        }
        prof_collect(c, marks);
        ++batch_iterations;
        ip = 0;                                                              // mod.rs:223
        for (int i = 0; i < B; ++i) {
            if (ret[i] >= 0) continue;
            const StatusRec& s = c->status_host[i];
            if (s.potrf_info != 0 || (s.flags & FLAG_NAN_PQ)) ret[i] = LPIPM_NUMERICAL_PROBLEM;   // mod.rs:215
            else if (s.status == ST_OPTIMAL) ret[i] = LPIPM_OK;              // mod.rs:231
            else if (s.status == ST_INFEASIBLE) ret[i] = LPIPM_INFEASIBLE;   // :232
            else if (s.status == ST_UNBOUNDED) ret[i] = LPIPM_UNBOUNDED;     // :233
            if (ret[i] >= 0) { its[i] = iteration; --running; }
        }
        // the refinement launches of the next iteration are enqueued if ANY running member asks for them; which members
        // they touch is each member's own device word (k_scalar_indicators), the same decision as when it is solved alone
        c->refine_now = c->refine == 2;
        for (int i = 0; i < B && c->refine == 1 && !c->refine_now; ++i)
            if (ret[i] < 0 && c->status_host[i].rho_mu <= refine_below()) c->refine_now = true;
    }
    for (int i = 0; i < B; ++i)
        if (ret[i] < 0) { ret[i] = LPIPM_ITERATION_LIMIT; its[i] = o->max_iter; }   // mod.rs:237-239
    LP_TRY(vec_final_x(v, c->xout, st, nullptr));                            // mod.rs:231/238, :165 (every LP)
    LP_TRY(copy_status(c));
    for (int i = 0; i < B; ++i) {     // rows[i]: member i's row in the caller's numbering (identity when null)
        if (ret[i] != LPIPM_OK && ret[i] != LPIPM_ITERATION_LIMIT) continue;
        const uint64_t row = rows ? rows[i] : (uint64_t)i;
        const char* src = (const char*)c->xout + (size_t)(c->bt.first + i) * c->bstride;
        if (xo.dev) LP_HIP(hipMemcpyAsync(xo.dev + row * xo.stride_bytes, src, c->n * sizeof(double), hipMemcpyDeviceToDevice, st));
        else if (xo.host[row]) LP_HIP(hipMemcpyAsync(xo.host[row], src, c->n * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (c->profiling) LP_HIP(hipEventRecord(c->ev_end, st));
    LP_HIP(hipStreamSynchronize(st));
    if (c->profiling) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->ev_begin, c->ev_end);
        c->times.total_ms = ms;
        c->times.adat_ms = c->tag_ms[T_ADAT]; c->times.potrf_ms = c->tag_ms[T_POTRF];
        c->times.trsv_ms = c->tag_ms[T_TRSV]; c->times.gemv_ms = c->tag_ms[T_GEMV]; c->times.vec_ms = c->tag_ms[T_VEC];
        c->times.adat_launches = batch_iterations;     // launches of the batched kernel (each covers all B members)
        c->times.iterations = batch_iterations;        // lockstep iterations of the batch (= the slowest member's count)
        c->times.gemv_passes = c->gemv_passes;
    }
    for (int i = 0; i < B; ++i) {
        status_out[i] = ret[i];
        const bool has_x = ret[i] == LPIPM_OK || ret[i] == LPIPM_ITERATION_LIMIT;
        if (fun_out) fun_out[i] = has_x ? c->status_host[i].obj : NAN;
        if (its_out) its_out[i] = its[i];
    }
    return LPIPM_OK;
}

extern "C" int lpipm_upload_lockstep(lpipm_ctx* c, uint64_t count, uint64_t m, uint64_t n, const double* const* A,
                                     const double* const* b, const double* const* cc, const double* c0) {
    if (count < 1 || count > 4096) return LPIPM_ERR_BAD_ARGUMENT;
    return upload_impl(c, (int)count, m, n, A, n, b, cc, c0, 0);
}
extern "C" int lpipm_solve_lockstep(lpipm_ctx* c, const lpipm_opts* o, double* const* x_slack_out, double* fun_out,
                                    uint64_t* iterations_out, int32_t* status_out) {
    XOut xo; xo.host = x_slack_out;
    return solve_lockstep(c, o, xo, nullptr, fun_out, iterations_out, status_out);
}
extern "C" int lpipm_solve_lockstep_device(lpipm_ctx* c, const lpipm_opts* o, void* x_dev_out, uint64_t row_stride,
                                           double* fun_out, uint64_t* iterations_out, int32_t* status_out) {
    if (!c || !x_dev_out || row_stride < c->n) return LPIPM_ERR_BAD_ARGUMENT;
    XOut xo; xo.dev = (char*)x_dev_out; xo.stride_bytes = row_stride * sizeof(double);
    return solve_lockstep(c, o, xo, nullptr, fun_out, iterations_out, status_out);
}

// Bytes one member of a lockstep batch of this shape occupies: the real layout (a measuring pass of layout_problem on a
// scratch context carrying only the geometry), not a formula that drifts from it.
static size_t lockstep_bytes_per_lp(const lpipm_ctx* c, uint64_t m, uint64_t n) {
    lpipm_ctx t;
    t.num_cu = c->num_cu; t.refine = c->refine; t.B = 32;
    t.mp = (int)round_up(m, NB); t.np = (int)round_up(n, BK); t.npa = t.np;
    t.nsplit = t.mp / GEMVT_ROWS;
    t.units_env = c->units_env;
    plan_adat(&t, t.B);
    Arena measure;
    if (layout_problem(&t, measure, false) != LPIPM_OK) return (size_t)-1;
    return (size_t)round_up(measure.off, 4096);
}

// A shard of independent LPs on one device.
//  1. Members of one shape (>= 2 of them, Cholesky arm) are solved as lockstep batches, in chunks that fit
//     the memory budget: one launch per kernel for the whole chunk.
//  2. The rest (odd shapes, QR arms) are latency-bound one by one (the factorisation's diagonal chain
//     keeps 1 of 256 CUs busy), so `batch_concurrency` contexts -- each with its own stream and buffers,
//     each driven by its own host thread -- solve different members at the same time; independent
//     streams need no cross-stream synchronisation.  Members are handed out through an atomic counter.
// Every member's result depends only on its own inputs.
static int batch_impl(lpipm_ctx* c, uint64_t count, const uint64_t* m, const uint64_t* n,
                      const double* const* A, const double* const* b, const double* const* cc,
                      const double* c0, const lpipm_opts* o, const XOut& xo,
                      double* fun_out, uint64_t* iterations_out, int32_t* status_out) {
    if (!c || !o || (count && (!m || !n || !A || !b || !cc || !xo.valid() || !status_out)))
        return LPIPM_ERR_BAD_ARGUMENT;
    std::vector<uint64_t> rest;                      // members left to the one-by-one path
    if (c->lockstep_max != 0 && o->solver_type == LPIPM_SOLVER_CHOLESKY) {
        LP_HIP(hipSetDevice(c->device));
        std::vector<char> taken(count, 0);
        for (uint64_t i = 0; i < count; ++i) {
            if (taken[i]) continue;
            std::vector<uint64_t> grp;
            for (uint64_t j = i; j < count; ++j)
                if (!taken[j] && m[j] == m[i] && n[j] == n[i]) grp.push_back(j);
            if (grp.size() < 2) continue;
            for (uint64_t j : grp) taken[j] = 1;
            // chunk size: the configured maximum, and what fits in ~60 % of the free memory (two chunks are
            // resident: one being solved, the next one being uploaded)
            size_t free_b = 0, total_b = 0;
            LP_HIP(hipMemGetInfo(&free_b, &total_b));
            free_b += c->arena_bytes;                // the current arena is released before the next one is made
            const double per_lp = (double)lockstep_bytes_per_lp(c, m[i], n[i]);     // the real arena layout of one member
            size_t chunk;
            if (c->lockstep_max > 0) chunk = (size_t)c->lockstep_max;
            else if (grp.size() > 32) chunk = 32;
            else if (grp.size() >= 16) chunk = (grp.size() + 1) / 2;   // two chunks: the second upload hides behind the first solve
            else chunk = grp.size();
            const size_t fit = (size_t)(0.3 * (double)free_b / per_lp);
            if (chunk > fit) chunk = fit;
            if (chunk < 2) { for (uint64_t j : grp) rest.push_back(j); continue; }
            // chunks of the group; a last chunk of one member goes to the one-by-one path
            struct Chunk { size_t k0, g; std::vector<const double*> A, b, c; std::vector<double> c0; };
            std::vector<Chunk> chunks;
            for (size_t k0 = 0; k0 < grp.size(); k0 += chunk) {
                const size_t g = (k0 + chunk < grp.size() ? k0 + chunk : grp.size()) - k0;
                if (g < 2) { rest.push_back(grp[k0]); continue; }
                Chunk ch{k0, g, std::vector<const double*>(g), std::vector<const double*>(g), std::vector<const double*>(g),
                         std::vector<double>(g, 0.0)};
                for (size_t k = 0; k < g; ++k) {
                    const uint64_t j = grp[k0 + k];
                    ch.A[k] = A[j]; ch.b[k] = b[j]; ch.c[k] = cc[j];
                    if (c0) ch.c0[k] = c0[j];
                }
                chunks.push_back(std::move(ch));
            }
            // Pipeline over two contexts: while chunk q is being solved on one, a helper thread uploads
            // chunk q+1 into the other (host staging + copy engine vs. compute: the PCIe time of a large
            // batch hides behind the solves).
            lpipm_ctx* pipe[2] = {c, nullptr};
            if (chunks.size() > 1) {
                if (c->workers.empty()) {
                    lpipm_ctx* w = nullptr;
                    const int rcw = lpipm_create(c->device, &w);
                    if (rcw != LPIPM_OK) return rcw;
                    c->workers.push_back(w);
                }
                pipe[1] = c->workers[0];
            }
            auto upload_chunk = [&](lpipm_ctx* w, const Chunk& ch) -> int {
                (void)hipSetDevice(w->device);
                return upload_impl(w, (int)ch.g, m[i], n[i], ch.A.data(), n[i], ch.b.data(), ch.c.data(), ch.c0.data(), 0);
            };
            int rc_up = chunks.empty() ? LPIPM_OK : upload_chunk(pipe[0], chunks[0]);
            for (size_t q = 0; q < chunks.size(); ++q) {
                const Chunk& ch = chunks[q];
                lpipm_ctx* w = pipe[q & 1];
                int rc_next = LPIPM_OK;
                std::thread up;
                if (q + 1 < chunks.size()) up = std::thread([&, q] { rc_next = upload_chunk(pipe[(q + 1) & 1], chunks[q + 1]); });
                const size_t g = ch.g;
                std::vector<double> gfun(g, NAN);
                std::vector<uint64_t> gits(g, 0), grow(g);
                std::vector<int32_t> gst(g, 0);
                for (size_t k = 0; k < g; ++k) grow[k] = grp[ch.k0 + k];
                int rc = rc_up;
                if (rc == LPIPM_OK) rc = solve_lockstep(w, o, xo, grow.data(), gfun.data(), gits.data(), gst.data());
                if (up.joinable()) up.join();
                if (rc >= 100) return rc;            // runtime failure: nothing sensible to continue with
                for (size_t k = 0; k < g; ++k) {
                    const uint64_t j = grp[ch.k0 + k];
                    status_out[j] = rc == LPIPM_OK ? gst[k] : rc;     // e.g. Unconstrained / InvalidParameter for all
                    if (fun_out) fun_out[j] = rc == LPIPM_OK ? gfun[k] : NAN;
                    if (iterations_out) iterations_out[j] = rc == LPIPM_OK ? gits[k] : 0;
                }
                rc_up = rc_next;
            }
        }
        for (uint64_t i = 0; i < count; ++i)
            if (!taken[i]) rest.push_back(i);
    } else {
        for (uint64_t i = 0; i < count; ++i) rest.push_back(i);
    }
    if (rest.empty()) return LPIPM_OK;
    int nworkers = c->batch_concurrency;
    if (nworkers == 0) {   // auto: latency-bound sizes gain ~3x from 4-8 members in flight (measured at
                           // 1024x2048: 165 -> 513 LP/s); sizes that fill the chip by themselves do not
        uint64_t mmax = 0;
        for (uint64_t i : rest) mmax = m[i] > mmax ? m[i] : mmax;
        nworkers = mmax <= 2048 ? 8 : 2;
    }
    if ((size_t)nworkers > rest.size()) nworkers = (int)rest.size();
    if (nworkers < 1) nworkers = 1;
    while ((int)c->workers.size() < nworkers - 1) {
        lpipm_ctx* w = nullptr;
        const int rc = lpipm_create(c->device, &w);
        if (rc != LPIPM_OK) return rc;
        c->workers.push_back(w);
    }
    std::atomic<uint64_t> next{0};
    std::atomic<int> fatal{LPIPM_OK};
    auto run = [&](lpipm_ctx* w) {
        (void)hipSetDevice(w->device);
        lpipm_opts opts = *o;
        opts.disp = 0;   // interleaved tables from concurrent members would be unreadable
        for (;;) {
            const uint64_t k = next.fetch_add(1);
            if (k >= rest.size() || fatal.load() != LPIPM_OK) break;
            const uint64_t i = rest[k];
            int rc = lpipm_upload(w, m[i], n[i], A[i], n[i], b[i], cc[i], c0 ? c0[i] : 0.0);
            double fun = NAN;
            uint64_t it = 0;
            if (rc == LPIPM_OK)
                rc = xo.dev ? solve_impl(w, &opts, nullptr, xo.dev + i * xo.stride_bytes, &fun, &it, nullptr)
                            : lpipm_solve(w, &opts, xo.host[i], &fun, &it, nullptr);
            status_out[i] = rc;
            if (fun_out) fun_out[i] = fun;
            if (iterations_out) iterations_out[i] = it;
            if (rc >= 100) { int expected = LPIPM_OK; fatal.compare_exchange_strong(expected, rc); }
        }
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < nworkers; ++t) threads.emplace_back(run, c->workers[t - 1]);
    run(c);
    for (std::thread& t : threads) t.join();
    return fatal.load();
}

extern "C" int lpipm_solve_batch(lpipm_ctx* c, uint64_t count, const uint64_t* m, const uint64_t* n,
                                 const double* const* A, const double* const* b, const double* const* cc,
                                 const double* c0, const lpipm_opts* o, double* const* x_slack_out,
                                 double* fun_out, uint64_t* iterations_out, int32_t* status_out) {
    XOut xo; xo.host = x_slack_out;
    return batch_impl(c, count, m, n, A, b, cc, c0, o, xo, fun_out, iterations_out, status_out);
}
extern "C" int lpipm_solve_batch_device(lpipm_ctx* c, uint64_t count, const uint64_t* m, const uint64_t* n,
                                        const double* const* A, const double* const* b, const double* const* cc,
                                        const double* c0, const lpipm_opts* o, void* x_dev_out, uint64_t row_stride,
                                        double* fun_out, uint64_t* iterations_out, int32_t* status_out) {
    if (count && !x_dev_out) return LPIPM_ERR_BAD_ARGUMENT;
    for (uint64_t i = 0; i < count && n; ++i)
        if (n[i] > row_stride) return LPIPM_ERR_BAD_ARGUMENT;
    XOut xo; xo.dev = (char*)x_dev_out; xo.stride_bytes = row_stride * sizeof(double);
    return batch_impl(c, count, m, n, A, b, cc, c0, o, xo, fun_out, iterations_out, status_out);
}

extern "C" int lpipm_set_batch_lockstep(lpipm_ctx* c, int max_group) {
    if (!c || max_group < -1 || max_group > 4096) return LPIPM_ERR_BAD_ARGUMENT;
    c->lockstep_max = max_group;
    return LPIPM_OK;
}

extern "C" int lpipm_set_batch_concurrency(lpipm_ctx* c, int nworkers) {
    if (!c || nworkers < 0 || nworkers > 64) return LPIPM_ERR_BAD_ARGUMENT;
    c->batch_concurrency = nworkers;
    return LPIPM_OK;
}

extern "C" int lpipm_set_collective(lpipm_ctx* c, int rank, int world, lpipm_allreduce_fn fn, void* user) {
    if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return LPIPM_ERR_BAD_ARGUMENT;
    c->rank = rank; c->world = world; c->coll = fn; c->coll_user = user;
    return LPIPM_OK;
}

extern "C" int lpipm_set_collective_on_stream(lpipm_ctx* c, int on) {
    if (!c) return LPIPM_ERR_BAD_ARGUMENT;
    c->coll_on_stream = on != 0;
    return LPIPM_OK;
}

extern "C" int lpipm_upload_nsplit(lpipm_ctx* c, uint64_t m, uint64_t n_total, uint64_t n_local, const double* A_local,
                                   uint64_t lda, const double* b, const double* c_local, double c0) {
    if (!c || n_local == 0 || n_local > n_total) return LPIPM_ERR_BAD_ARGUMENT;
    const int rc = lpipm_upload_slack(c, m, n_local, A_local, lda, b, c_local, c0, 0);
    if (rc != LPIPM_OK) return rc;
    if (c->world > 1) {
        const size_t need = (size_t)c->mp * ((size_t)c->mp + 128) / 2;
        if (need != c->mpack_count) {
            if (c->mpack) { LP_HIP(hipFree(c->mpack)); c->mpack = nullptr; c->mpack_count = 0; }
            LP_HIP(hipMalloc((void**)&c->mpack, need * sizeof(double)));
            c->mpack_count = need;
        }
    }
    if (c->world > 1 && !c->st_c) {
        if (hipStreamCreateWithFlags(&c->st_c, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_c0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_c1, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (c->st_c) (void)hipStreamDestroy(c->st_c);
            c->st_c = nullptr;                   // M is then reduced in one block after the launch
        }
    }
    c->colsplit = true;
    c->va.n_total = (long long)n_total;
    c->va.gs = c->gs;
    return LPIPM_OK;
}

extern "C" int lpipm_set_profiling(lpipm_ctx* c, int on) {
    if (!c) return LPIPM_ERR_BAD_ARGUMENT;
    c->profiling = on < 0 ? 0 : (on > 2 ? 1 : on);
    return LPIPM_OK;
}
extern "C" int lpipm_get_phase_times(const lpipm_ctx* c, lpipm_phase_times* out) {
    if (!c || !out) return LPIPM_ERR_BAD_ARGUMENT;
    *out = c->times;
    return LPIPM_OK;
}

// ------------------------------------------------------------------------------------------------
// kernel-granularity entry points (parity tests / micro-benchmarks)
template <typename F>
static int timed_repeats(lpipm_ctx* c, int repeats, double* ms_out, F&& body) {
    if (repeats < 1) repeats = 1;
    float total = 0.f;
    for (int r = 0; r < repeats; ++r) {
        LP_HIP(hipEventRecord(c->ev_begin, c->st));
        LP_TRY(body());
        LP_HIP(hipEventRecord(c->ev_end, c->st));
        LP_HIP(hipStreamSynchronize(c->st));
        float ms = 0.f;
        LP_HIP(hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
        total += ms;
    }
    if (ms_out) *ms_out = total / repeats;
    return LPIPM_OK;
}

extern "C" int lpipm_k_adat(lpipm_ctx* c, const double* dinv, double* M_out, int repeats, double* ms_out) {
    if (!c || !dinv || !M_out) return LPIPM_ERR_BAD_ARGUMENT;
    if (!c->has_problem) return LPIPM_ERR_NO_PROBLEM;
    LP_HIP(hipSetDevice(c->device));
    LP_HIP(hipMemcpyAsync(c->va.dinv, dinv, c->n * sizeof(double), hipMemcpyHostToDevice, c->st));
    LP_TRY(timed_repeats(c, repeats, ms_out, [&]() -> int { LP_HIP(run_adat(c, Batch{})); return LPIPM_OK; }));
    LP_HIP(hipMemcpy2DAsync(M_out, c->m * sizeof(double), c->M, (size_t)c->mp * sizeof(double),
                            c->m * sizeof(double), c->m, hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipStreamSynchronize(c->st));
    return LPIPM_OK;
}

static int kbuf_ensure(lpipm_ctx* c, int mp) {
    if (c->kmp == mp) return LPIPM_OK;
    LP_HIP(hipStreamSynchronize(c->st));
    free_list(c->kallocs);
    factor_plan_destroy(c->kplan);
    c->kmp = 0;
    LP_TRY(dalloc(c->kallocs, nullptr, &c->kM, (size_t)mp * mp, c->st));
    LP_TRY(dalloc(c->kallocs, nullptr, &c->kM0, (size_t)mp * mp, c->st));
    LP_TRY(dalloc(c->kallocs, nullptr, &c->kR, (size_t)2 * mp, c->st));
    LP_TRY(dalloc(c->kallocs, nullptr, &c->kY, (size_t)2 * mp, c->st));
    LP_TRY(dalloc(c->kallocs, nullptr, &c->kinfo, 1, c->st));
    LP_TRY(dalloc(c->kallocs, nullptr, &c->ktau, (size_t)mp, c->st));
    Arena measure;
    LP_HIP(factor_plan_create(c->kplan, c->kM, mp, mp, measure, false, c->st, super_for(mp), merge_edge_for(1)));
    char* kar = nullptr;
    LP_TRY(dalloc(c->kallocs, nullptr, &kar, measure.off + 256, c->st));   // zeroed
    Arena real;
    real.base = kar;
    LP_HIP(factor_plan_create(c->kplan, c->kM, mp, mp, real, true, c->st, super_for(mp), merge_edge_for(1)));
    c->kmp = mp;
    return LPIPM_OK;
}

extern "C" int lpipm_k_potrf(lpipm_ctx* c, uint64_t m, double* M_inout, int32_t* info_out, int repeats,
                             double* ms_out) {
    if (!c || !M_inout || m == 0 || m > (1u << 20)) return LPIPM_ERR_BAD_ARGUMENT;
    LP_HIP(hipSetDevice(c->device));
    const int mp = (int)round_up(m, NB);
    LP_TRY(kbuf_ensure(c, mp));
    // padded pristine copy: [[M, 0], [0, I]]
    std::vector<double> pad((size_t)(mp - m), 1.0);
    LP_HIP(hipMemsetAsync(c->kM0, 0, (size_t)mp * mp * sizeof(double), c->st));
    LP_HIP(hipMemcpy2DAsync(c->kM0, (size_t)mp * sizeof(double), M_inout, m * sizeof(double), m * sizeof(double),
                            m, hipMemcpyHostToDevice, c->st));
    if (mp > (int)m)
        LP_HIP(hipMemcpy2DAsync(c->kM0 + m * mp + m, (size_t)(mp + 1) * sizeof(double), pad.data(), sizeof(double),
                                sizeof(double), mp - m, hipMemcpyHostToDevice, c->st));
    LP_HIP(hipStreamSynchronize(c->st));
    if (repeats < 1) repeats = 1;
    float total = 0.f;
    for (int r = 0; r < repeats; ++r) {
        LP_HIP(hipMemcpyAsync(c->kM, c->kM0, (size_t)mp * mp * sizeof(double), hipMemcpyDeviceToDevice, c->st));
        LP_HIP(hipEventRecord(c->ev_begin, c->st));
        LP_HIP(launch_potrf(c->kM, mp, mp, c->kplan, c->kinfo, c->st, Batch{}, lookahead(c)));
        LP_HIP(hipEventRecord(c->ev_end, c->st));
        LP_HIP(hipStreamSynchronize(c->st));
        float ms = 0.f;
        LP_HIP(hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
        total += ms;
    }
    if (ms_out) *ms_out = total / repeats;
    if (lp_knob("LPIPM_DIAG_STAMPS")) {  // debug aid: cycle stamps of the first diagonal-block kernel
        long long* d = nullptr; long long h[64] = {0};
        if (hipMalloc((void**)&d, sizeof(h)) == hipSuccess) {
            g_diag_stamps = d;
            (void)hipMemset(d, 0, sizeof(h));
            (void)hipMemcpyAsync(c->kM, c->kM0, (size_t)mp * mp * sizeof(double), hipMemcpyDeviceToDevice, c->st);
            (void)launch_potrf(c->kM, mp, mp, c->kplan, c->kinfo, c->st);
            (void)hipStreamSynchronize(c->st);
            g_diag_stamps = nullptr;
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            (void)hipFree(d);
            fprintf(stderr, "diag stamps (cycles): E(0) %lld, P tile + wait for all eight on wave 0 %lld, barrier to barrier (k = 1) %lld, "
                    "all block columns %lld, last stores %lld\n  k = 1, cycles after the barrier, per wave: P tile done",
                    h[1]-h[0], h[2]-h[1], h[3]-h[1], h[6]-h[0], h[7]-h[6]);
            for (int w = 0; w < 16; ++w) fprintf(stderr, " %lld", h[32 + w] - h[1]);
            fprintf(stderr, "\n  at the next barrier");
            for (int w = 0; w < 16; ++w) fprintf(stderr, " %lld", h[16 + w] - h[1]);
            fprintf(stderr, "\n");
        }
    }
    int32_t info = 0;
    LP_HIP(hipMemcpyAsync(&info, c->kinfo, sizeof(int32_t), hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipMemcpy2DAsync(M_inout, m * sizeof(double), c->kM, (size_t)mp * sizeof(double), m * sizeof(double), m,
                            hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipStreamSynchronize(c->st));
    if (info_out) *info_out = info;
    c->kchol_valid = true;
    return LPIPM_OK;
}

extern "C" int lpipm_k_chol_solve(lpipm_ctx* c, uint64_t m, int nrhs, const double* R, double* V, int repeats,
                                  double* ms_out) {
    if (!c || !R || !V || (nrhs != 1 && nrhs != 2)) return LPIPM_ERR_BAD_ARGUMENT;
    const int mp = (int)round_up(m, NB);
    if (c->kmp != mp || !c->kchol_valid) return LPIPM_ERR_NO_PROBLEM;  // needs a preceding lpipm_k_potrf of this size
    LP_HIP(hipSetDevice(c->device));
    if (repeats < 1) repeats = 1;
    float total = 0.f;
    for (int r = 0; r < repeats; ++r) {
        LP_HIP(hipMemsetAsync(c->kR, 0, (size_t)2 * mp * sizeof(double), c->st));
        LP_HIP(hipMemcpy2DAsync(c->kR, (size_t)mp * sizeof(double), R, m * sizeof(double), m * sizeof(double), nrhs,
                                hipMemcpyHostToDevice, c->st));
        LP_HIP(hipEventRecord(c->ev_begin, c->st));
        LP_HIP(launch_chol_solve(c->kM, mp, c->kplan, nrhs, c->kR, c->kY, c->st));
        LP_HIP(hipEventRecord(c->ev_end, c->st));
        LP_HIP(hipStreamSynchronize(c->st));
        float ms = 0.f;
        LP_HIP(hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
        total += ms;
    }
    if (ms_out) *ms_out = total / repeats;
    LP_HIP(hipMemcpy2DAsync(V, m * sizeof(double), c->kR, (size_t)mp * sizeof(double), m * sizeof(double), nrhs,
                            hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipStreamSynchronize(c->st));
    return LPIPM_OK;
}

extern "C" int lpipm_k_symv_residual(lpipm_ctx* c, uint64_t m, const double* M, int nrhs, const double* V, const double* R0,
                                     double* Rho) {
    if (!c || !M || !V || !R0 || !Rho || m == 0 || m > 16384 || (nrhs != 1 && nrhs != 2)) return LPIPM_ERR_BAD_ARGUMENT;
    LP_HIP(hipSetDevice(c->device));
    const int mp = (int)round_up(m, NB);
    LP_TRY(kbuf_ensure(c, mp));
    double *ws = nullptr, *vbuf = nullptr;
    // every failure leaves through the one exit below, which frees both buffers
    hipError_t e = hipMalloc((void**)&ws, symv_slab_doubles(mp) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&vbuf, (size_t)6 * mp * sizeof(double));
    if (e == hipSuccess) e = hipMemsetAsync(vbuf, 0, (size_t)6 * mp * sizeof(double), c->st);
    if (e == hipSuccess) e = hipMemsetAsync(c->kM0, 0, (size_t)mp * mp * sizeof(double), c->st);
    if (e == hipSuccess) e = hipMemcpy2DAsync(c->kM0, (size_t)mp * sizeof(double), M, m * sizeof(double), m * sizeof(double), m, hipMemcpyHostToDevice, c->st);
    if (e == hipSuccess) e = hipMemcpy2DAsync(vbuf, (size_t)mp * sizeof(double), V, m * sizeof(double), m * sizeof(double), nrhs, hipMemcpyHostToDevice, c->st);
    if (e == hipSuccess) e = hipMemcpy2DAsync(vbuf + 2 * mp, (size_t)mp * sizeof(double), R0, m * sizeof(double), m * sizeof(double), nrhs, hipMemcpyHostToDevice, c->st);
    if (e == hipSuccess) e = launch_symv_residual(c->kM0, mp, mp, nrhs, vbuf, mp, vbuf + 2 * mp, mp, vbuf + 4 * mp, mp, ws, c->st);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(Rho, m * sizeof(double), vbuf + 4 * mp, (size_t)mp * sizeof(double), m * sizeof(double), nrhs, hipMemcpyDeviceToHost, c->st);
    const hipError_t es = hipStreamSynchronize(c->st);      // also on failure: nothing may still use the buffers freed next
    if (e == hipSuccess) e = es;
    if (ws) (void)hipFree(ws);
    if (vbuf) (void)hipFree(vbuf);
    LP_HIP(e);
    return LPIPM_OK;
}

extern "C" int lpipm_k_qr_solve(lpipm_ctx* c, uint64_t m, const double* M, int nrhs, const double* R, double* V,
                                int32_t* info_out, double* ms_out) {
    if (!c || !M || !R || !V || m == 0 || m > 16384 || (nrhs != 1 && nrhs != 2)) return LPIPM_ERR_BAD_ARGUMENT;
    LP_HIP(hipSetDevice(c->device));
    const int mp = (int)round_up(m, NB);
    LP_TRY(kbuf_ensure(c, mp));
    std::vector<double> pad((size_t)(mp - m), 1.0);
    LP_HIP(hipMemsetAsync(c->kM, 0, (size_t)mp * mp * sizeof(double), c->st));
    LP_HIP(hipMemcpy2DAsync(c->kM, (size_t)mp * sizeof(double), M, m * sizeof(double), m * sizeof(double), m,
                            hipMemcpyHostToDevice, c->st));
    if (mp > (int)m)
        LP_HIP(hipMemcpy2DAsync(c->kM + m * mp + m, (size_t)(mp + 1) * sizeof(double), pad.data(), sizeof(double),
                                sizeof(double), mp - m, hipMemcpyHostToDevice, c->st));
    LP_HIP(hipMemsetAsync(c->kR, 0, (size_t)2 * mp * sizeof(double), c->st));
    LP_HIP(hipMemcpy2DAsync(c->kR, (size_t)mp * sizeof(double), R, m * sizeof(double), m * sizeof(double), nrhs,
                            hipMemcpyHostToDevice, c->st));
    LP_HIP(hipEventRecord(c->ev_begin, c->st));
    LP_HIP(launch_qr_factor(c->kM, mp, mp, c->ktau, c->kinfo, c->st));
    LP_HIP(launch_qr_solve(c->kM, mp, mp, c->ktau, nrhs, c->kR, c->kinfo, c->st));
    LP_HIP(hipEventRecord(c->ev_end, c->st));
    int32_t info = 0;
    LP_HIP(hipMemcpyAsync(&info, c->kinfo, sizeof(int32_t), hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipMemcpy2DAsync(V, m * sizeof(double), c->kR, (size_t)mp * sizeof(double), m * sizeof(double), nrhs,
                            hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipStreamSynchronize(c->st));
    float ms = 0.f;
    LP_HIP(hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
    if (ms_out) *ms_out = ms;
    if (info_out) *info_out = info;
    c->kchol_valid = false;   // kM no longer holds a Cholesky factor: lpipm_k_chol_solve needs a new lpipm_k_potrf
    return LPIPM_OK;
}

extern "C" int lpipm_k_gemv_n(lpipm_ctx* c, int nrhs, const double* W, double* Y, int repeats, double* ms_out) {
    if (!c || !W || !Y || (nrhs != 1 && nrhs != 2)) return LPIPM_ERR_BAD_ARGUMENT;
    if (!c->has_problem) return LPIPM_ERR_NO_PROBLEM;
    LP_HIP(hipSetDevice(c->device));
    LP_HIP(hipMemcpy2DAsync(c->va.W, (size_t)c->np * sizeof(double), W, c->n * sizeof(double), c->n * sizeof(double),
                            nrhs, hipMemcpyHostToDevice, c->st));
    LP_TRY(timed_repeats(c, repeats, ms_out, [&]() -> int {
        LP_HIP(ctx_gemv_n(c, nrhs, c->va.W, nullptr, nullptr, c->va.R, Batch{}));
        return LPIPM_OK;
    }));
    LP_HIP(hipMemcpy2DAsync(Y, c->m * sizeof(double), c->va.R, (size_t)c->mp * sizeof(double), c->m * sizeof(double),
                            nrhs, hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipStreamSynchronize(c->st));
    return LPIPM_OK;
}

extern "C" int lpipm_k_gemv_t(lpipm_ctx* c, int nrhs, const double* V, double* U, int repeats, double* ms_out) {
    if (!c || !V || !U || (nrhs != 1 && nrhs != 2)) return LPIPM_ERR_BAD_ARGUMENT;
    if (!c->has_problem) return LPIPM_ERR_NO_PROBLEM;
    LP_HIP(hipSetDevice(c->device));
    LP_HIP(hipMemsetAsync(c->va.R, 0, (size_t)2 * c->mp * sizeof(double), c->st));
    LP_HIP(hipMemcpy2DAsync(c->va.R, (size_t)c->mp * sizeof(double), V, c->m * sizeof(double), c->m * sizeof(double),
                            nrhs, hipMemcpyHostToDevice, c->st));
    LP_TRY(timed_repeats(c, repeats, ms_out, [&]() -> int {
        LP_HIP(ctx_gemv_t(c, nrhs, c->va.R, Batch{}));
        LP_HIP(launch_gemv_t_reduce(c->ATpart, c->nsplit, nrhs, c->np, c->va.W, c->np, c->st));
        return LPIPM_OK;
    }));
    LP_HIP(hipMemcpy2DAsync(U, c->n * sizeof(double), c->va.W, (size_t)c->np * sizeof(double), c->n * sizeof(double),
                            nrhs, hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipStreamSynchronize(c->st));
    return LPIPM_OK;
}

// One loop body of solve_normal_form (mod.rs:215-222) on the uploaded problem from a GIVEN iterate: what the loop does
// between two status read-backs -- residuals at the point (feasible_point.rs:122-125), normal equations, factor,
// predictor, corrector, step length, step.  For the differential tests of the vector stage (rhat.rs, delta.rs,
// feasible_point.rs:53-106) on arbitrary iterates, including ip = 1.
extern "C" int lpipm_k_iteration(lpipm_ctx* c, const lpipm_opts* o, int ip, double* x, double* y, double* z, double* tau,
                                 double* kappa, double* d_x, double* d_y, double* d_z, double* d_tk, double* alpha_out,
                                 int32_t* info_out) {
    if (!c || !o || !x || !y || !z || !tau || !kappa || !d_x || !d_y || !d_z || !d_tk || !alpha_out) return LPIPM_ERR_BAD_ARGUMENT;
    if (!c->has_problem) return LPIPM_ERR_NO_PROBLEM;
    if (c->B != 1 || c->colsplit) return LPIPM_ERR_UNSUPPORTED;
    LP_HIP(hipSetDevice(c->device));
    VecArgs& v = c->va;
    hipStream_t st = c->st;
    c->refine_now = c->refine == 2;
    c->factor_in_head = false;
    vec_blind_start(v, st);                                   // clears done / flags; the iterate is overwritten next
    LP_HIP(hipMemcpyAsync(v.x, x, c->n * sizeof(double), hipMemcpyHostToDevice, st));
    LP_HIP(hipMemcpyAsync(v.y, y, c->m * sizeof(double), hipMemcpyHostToDevice, st));
    LP_HIP(hipMemcpyAsync(v.z, z, c->n * sizeof(double), hipMemcpyHostToDevice, st));
    const double tk[2] = {*tau, *kappa};
    LP_HIP(hipMemcpyAsync(v.S + S_TAU, &tk[0], sizeof(double), hipMemcpyHostToDevice, st));
    LP_HIP(hipMemcpyAsync(v.S + S_KAPPA, &tk[1], sizeof(double), hipMemcpyHostToDevice, st));
    LP_TRY(enqueue_residuals(c, 1, ip ? 1 : 0, o->tol));      // r_P, r_D, r_G, mu at the point
    LP_TRY(enqueue_iteration(c, ip ? 1 : 0, o));
    double sc[S_COUNT];
    LP_HIP(hipMemcpyAsync(sc, v.S, sizeof(sc), hipMemcpyDeviceToHost, st));
    LP_HIP(hipMemcpyAsync(x, v.x, c->n * sizeof(double), hipMemcpyDeviceToHost, st));
    LP_HIP(hipMemcpyAsync(y, v.y, c->m * sizeof(double), hipMemcpyDeviceToHost, st));
    LP_HIP(hipMemcpyAsync(z, v.z, c->n * sizeof(double), hipMemcpyDeviceToHost, st));
    LP_HIP(hipMemcpyAsync(d_x, v.dx, c->n * sizeof(double), hipMemcpyDeviceToHost, st));
    LP_HIP(hipMemcpyAsync(d_y, v.dy, c->m * sizeof(double), hipMemcpyDeviceToHost, st));
    LP_HIP(hipMemcpyAsync(d_z, v.dz, c->n * sizeof(double), hipMemcpyDeviceToHost, st));
    LP_HIP(hipStreamSynchronize(st));
    *tau = sc[S_TAU]; *kappa = sc[S_KAPPA]; d_tk[0] = sc[S_DTAU]; d_tk[1] = sc[S_DKAPPA]; *alpha_out = sc[S_ALPHA];
    if (info_out) *info_out = c->status_host->potrf_info;
    return LPIPM_OK;
}

extern "C" int lpipm_k_gemv_dual(lpipm_ctx* c, const double* w, const double* v, double* Aw_out, double* ATv_out, int repeats,
                                 double* ms_out) {
    if (!c || !w || !v || !Aw_out || !ATv_out) return LPIPM_ERR_BAD_ARGUMENT;
    if (!c->has_problem) return LPIPM_ERR_NO_PROBLEM;
    LP_HIP(hipSetDevice(c->device));
    VecArgs& va = c->va;
    LP_HIP(hipMemsetAsync(va.W, 0, (size_t)c->np * sizeof(double), c->st));
    LP_HIP(hipMemsetAsync(va.R, 0, (size_t)c->mp * sizeof(double), c->st));
    LP_HIP(hipMemcpyAsync(va.W, w, c->n * sizeof(double), hipMemcpyHostToDevice, c->st));
    LP_HIP(hipMemcpyAsync(va.R, v, c->m * sizeof(double), hipMemcpyHostToDevice, c->st));
    const int nch = gemv_dual_chunks(c->npa);
    LP_TRY(timed_repeats(c, repeats, ms_out, [&]() -> int {
        LP_HIP(launch_gemv_dual(c->A, c->npa, c->mp, c->npa, va.W, va.R, va.Ax, c->ATpart, c->np, c->st));
        LP_HIP(launch_slack_n(c->ns, c->nx, 1, va.W, c->np, va.Ax, c->mp, c->st));
        LP_HIP(launch_slack_t(c->ns, c->nx, 1, c->nsplit, va.R, c->mp, c->ATpart, c->np, c->st));
        return LPIPM_OK;
    }));
    // the consumers' folds, on the host: chunk slabs of A.w, row-block slabs of A^T.v, in index order
    std::vector<double> ax((size_t)nch * c->mp), at((size_t)c->nsplit * c->np);
    LP_HIP(hipMemcpyAsync(ax.data(), va.Ax, ax.size() * sizeof(double), hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipMemcpyAsync(at.data(), c->ATpart, at.size() * sizeof(double), hipMemcpyDeviceToHost, c->st));
    LP_HIP(hipStreamSynchronize(c->st));
    for (uint64_t i = 0; i < c->m; ++i) { double s = 0.0; for (int ch = 0; ch < nch; ++ch) s += ax[(size_t)ch * c->mp + i]; Aw_out[i] = s; }
    for (uint64_t j = 0; j < c->n; ++j) { double s = 0.0; for (int sp = 0; sp < c->nsplit; ++sp) s += at[(size_t)sp * c->np + j]; ATv_out[j] = s; }
    return LPIPM_OK;
}

extern "C" int lpipm_k_mfma_f64_probe(lpipm_ctx* c, int iters, double* tflops_out, double* ms_out) {
    if (!c || iters < 1) return LPIPM_ERR_BAD_ARGUMENT;
    LP_HIP(hipSetDevice(c->device));
    const int blocks = c->num_cu * 2;
    double* sink = nullptr;
    LP_HIP(hipMalloc((void**)&sink, (size_t)blocks * 256 * sizeof(double)));
    double ms = 0.0;
    int rc = timed_repeats(c, 3, &ms, [&]() -> int { LP_HIP(launch_mfma_probe(iters, sink, blocks, c->st)); return LPIPM_OK; });
    (void)hipFree(sink);
    if (rc != LPIPM_OK) return rc;
    // per wave and loop trip: 16 independent accumulators x one 16x16x4 MFMA = 16 * 2048 flop
    const double flop = (double)blocks * 4.0 * (double)iters * 16.0 * 2048.0;
    if (ms_out) *ms_out = ms;
    if (tflops_out) *tflops_out = flop / (ms * 1e-3) / 1e12;
    return LPIPM_OK;
}
