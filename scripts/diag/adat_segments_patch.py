"""Scratch-copy instrumentation: s_memtime stamps at the phase boundaries of the A.D.A^T k-loop (prio variant),
accumulated per wave 0 of each workgroup; prints the median split.  Perturbs the loop (~+10 %); shows where a wave's
time goes, not the absolute speed."""
p = 'lp_amd/csrc/kernels_gemm.hip'
s = open(p).read()
loop_old = s[s.index("    for (int kt = kb; kt < ke; ++kt) {\n        const bool more = kt + 1 < ke;"):s.index("// C tile <- beta*C + alpha*acc.")]
new_loop = r'''    unsigned long long seg[7] = {0, 0, 0, 0, 0, 0, 0};
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0)
    for (int kt = kb; kt < ke; ++kt) {
        const bool more = kt + 1 < ke;
        unsigned long long t0, t1, t2, t3, t4, t5, t6, t7;
        __builtin_amdgcn_sched_barrier(0); STAMP(t0);
        if (more) gload(kt + 1);
        d2 a[MTM], b[MTN];
#pragma unroll
        for (int mi = 0; mi < MTM; ++mi) a[mi] = *(const d2*)&ldsA[cur][wr * (16 * MTM) + mi * 16 + fr][fq * 2];
#pragma unroll
        for (int nj = 0; nj < MTN; ++nj) b[nj] = *(const d2*)&ldsB[cur][wc * (16 * MTN) + nj * 16 + fr][fq * 2];
        __builtin_amdgcn_sched_barrier(0); STAMP(t1);      // loads issued, fragments of round 0 in registers
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
                for (int nj = 0; nj < MTN; ++nj)
                    acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0); STAMP(t2);      // round-0 MFMAs issued
#pragma unroll
        for (int mi = 0; mi < MTM; ++mi) a[mi] = *(const d2*)&ldsA[cur][wr * (16 * MTM) + mi * 16 + fr][8 + fq * 2];
#pragma unroll
        for (int nj = 0; nj < MTN; ++nj) b[nj] = *(const d2*)&ldsB[cur][wc * (16 * MTN) + nj * 16 + fr][8 + fq * 2];
        __builtin_amdgcn_sched_barrier(0); STAMP(t3);      // fragments of round 1 in registers
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
                for (int nj = 0; nj < MTN; ++nj)
                    acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0); STAMP(t4);      // round-1 MFMAs issued
        __builtin_amdgcn_s_setprio(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0); STAMP(t5);      // prefetched k-tile has arrived
        if (more) lstore(cur ^ 1);
        __builtin_amdgcn_sched_barrier(0); STAMP(t6);      // ... and is in LDS
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0); STAMP(t7);
        seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2; seg[3] += t4 - t3; seg[4] += t5 - t4; seg[5] += t6 - t5; seg[6] += t7 - t6;
        cur ^= 1;
    }
    if (SCALE && threadIdx.x == 0 && ke - kb == 512 && blockIdx.x < 1024)
        for (int i = 0; i < 7; ++i) g_seg[8 * blockIdx.x + i] = seg[i];
#undef STAMP
}

'''
s = s.replace(loop_old, new_loop)
s = s.replace("template <bool SCALE, int MTM, int MTN>\n__device__ __forceinline__ void tile_mainloop(", "static __device__ unsigned long long g_seg[8 * 1024];\ntemplate <bool SCALE, int MTM, int MTN>\n__device__ __forceinline__ void tile_mainloop(", 1)
s = s.replace('#include "lpipm_internal.hpp"\n', '#include "lpipm_internal.hpp"\n#include <cstdio>\n#include <vector>\n#include <algorithm>\n', 1)
s = s.replace("hipError_t launch_gemm_grouped(", r'''void dbg_print_clock() {
    std::vector<unsigned long long> h(8 * 1024, 0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_seg), h.size() * sizeof(unsigned long long));
    const char* names[7] = {"issue prefetch + read round-0 fragments", "issue 32 MFMAs (round 0)", "read round-1 fragments", "issue 32 MFMAs (round 1)",
                            "wait for the prefetched k-tile", "scale + ds_write it", "barrier"};
    double tot = 0; double med[7];
    for (int i = 0; i < 7; ++i) {
        std::vector<double> v;
        for (int b = 0; b < 512; ++b) v.push_back((double)h[8 * b + i] / 512.0);
        std::sort(v.begin(), v.end());
        med[i] = v[v.size() / 2]; tot += med[i];
    }
    fprintf(stderr, "cycles per k-tile of wave 0 (median over 512 workgroups, instrumented loop): total %.0f\n", tot);
    for (int i = 0; i < 7; ++i) fprintf(stderr, "  %-42s %7.0f (%4.1f %%)\n", names[i], med[i], 100.0 * med[i] / tot);
}
hipError_t launch_gemm_grouped(''', 1)
open(p, 'w').write(s)
p = 'lp_amd/csrc/solver.hip'
s = open(p).read()
s = s.replace('extern "C" int lpipm_k_adat(', 'namespace lpipm { void dbg_print_clock(); }\nextern "C" int lpipm_k_adat(', 1)
s = s.replace("    LP_TRY(timed_repeats(c, repeats, ms_out, [&]() -> int { LP_HIP(run_adat(c, Batch{})); return LPIPM_OK; }));",
              "    LP_TRY(timed_repeats(c, repeats, ms_out, [&]() -> int { LP_HIP(run_adat(c, Batch{})); return LPIPM_OK; }));\n    lpipm::dbg_print_clock();")
open(p, 'w').write(s)
print("patched: segment stamps")
