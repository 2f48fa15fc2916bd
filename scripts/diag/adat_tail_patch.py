"""Scratch-copy diagnostic: when does each workgroup of the A.D.A^T launch finish its data-parallel tile, its stream-K
share, and the kernel (s_memrealtime, 100 MHz)?  Shows the tail: how long the chip waits for the slowest workgroups."""
p = 'lp_amd/csrc/kernels_gemm.hip'
s = open(p).read()
s = s.replace('#include "lpipm_internal.hpp"\n', '#include "lpipm_internal.hpp"\n#include <cstdio>\n#include <vector>\n#include <algorithm>\n', 1)
s = s.replace("template <bool SCALE>\n__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_streamk_w8_kernel(const GemmK p0) {\n    if (batch_done(p0.bk)) return;",
              "static __device__ unsigned long long g_tl[4 * 1024];\ntemplate <bool SCALE>\n__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_streamk_w8_kernel(const GemmK p0) {\n    if (batch_done(p0.bk)) return;\n    unsigned long long t_begin, t_dp, t_end;\n    asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_begin) :: \"memory\");", 1)
a = s.index("void gemm_nt_streamk_w8_kernel(const GemmK p0)")
i = s.index("    const long long total = (long long)(p.ntiles - ntiles_dp) * KT;", a)
s = s[:i] + "    asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_dp) :: \"memory\");\n" + s[i:]
j = s.index("        it += ke - kb;\n        first = false;\n    }\n}\n", i)
j2 = j + len("        it += ke - kb;\n        first = false;\n    }\n")
s = s[:j2] + "    asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_end) :: \"memory\");\n    if (SCALE && threadIdx.x == 0 && blockIdx.x < 1024) { g_tl[4 * blockIdx.x] = t_begin; g_tl[4 * blockIdx.x + 1] = t_dp; g_tl[4 * blockIdx.x + 2] = t_end; }\n" + s[j2:]
s = s.replace("hipError_t launch_gemm_grouped(", r'''void dbg_print_clock() {
    std::vector<unsigned long long> h(4 * 1024, 0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_tl), h.size() * sizeof(unsigned long long));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < 512; ++b) if (h[4 * b]) t0 = std::min(t0, h[4 * b]);
    std::vector<double> st, dp, en;
    for (int b = 0; b < 512; ++b) if (h[4 * b]) { st.push_back((h[4*b] - t0) * 0.01); dp.push_back((h[4*b+1] - t0) * 0.01); en.push_back((h[4*b+2] - t0) * 0.01); }
    std::sort(st.begin(), st.end()); std::sort(dp.begin(), dp.end()); std::sort(en.begin(), en.end());
    const size_t n = st.size();
    if (!n) { fprintf(stderr, "no stamps\n"); return; }
    fprintf(stderr, "A.D.A^T launch, %zu workgroups, us after the first one started: start median %.1f max %.1f | data-parallel tile done: min %.1f median %.1f p90 %.1f max %.1f | "
                    "stream-K share done: min %.1f median %.1f p90 %.1f max %.1f\n", n, st[n/2], st[n-1], dp[0], dp[n/2], dp[n*9/10], dp[n-1], en[0], en[n/2], en[n*9/10], en[n-1]);
}
hipError_t launch_gemm_grouped(''', 1)
open(p, 'w').write(s)
p = 'lp_amd/csrc/solver.hip'
s = open(p).read()
s = s.replace('extern "C" int lpipm_k_adat(', 'namespace lpipm { void dbg_print_clock(); }\nextern "C" int lpipm_k_adat(', 1)
old = "    LP_TRY(timed_repeats(c, repeats, ms_out, [&]() -> int { LP_HIP(run_adat(c, Batch{})); return LPIPM_OK; }));"
assert old in s
s = s.replace(old, old + "\n    lpipm::dbg_print_clock();")
open(p, 'w').write(s)
print("patched: tail diagnostic")
