"""Diagnostic build (apply to a SCRATCH copy, e.g. the GPU box's snapshot -- never commit the patched sources):
one s_memtime / s_memrealtime stamp pair around the k-loop of each workgroup's data-parallel tile of the A.D.A^T
kernel (8-wave form), written to a buffer of their own.  In-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz and
shader cycles per k-tile, as MI355X_MICROARCH.md 'DVFS give-back' item 6 prescribes (>= 2 s of back-to-back launches,
random data, median over workgroups).
Usage on the box:  python scripts/diag/adat_clock_patch.py && make -C lp_amd/csrc && python scripts/diag/adat_clock_run.py"""
p = 'lp_amd/csrc/kernels_gemm.hip'
s = open(p).read()
s = s.replace('#include "lpipm_internal.hpp"\n', '#include "lpipm_internal.hpp"\n#include <cstdio>\n#include <cstdlib>\n#include <vector>\n#include <algorithm>\n', 1)
s = s.replace("template <bool SCALE>\n__device__ __forceinline__ void tile_mainloop_w8(",
              "static __device__ unsigned long long g_clk[4 * 1024];\ntemplate <bool SCALE>\n__device__ __forceinline__ void tile_mainloop_w8(", 1)
a = s.index("__device__ __forceinline__ void tile_mainloop_w8(")
body0 = s.index("    d2 sa[2], sb[2], sv = (d2){1.0, 1.0};", a)
s = s[:body0] + ("    unsigned long long c0_, r0_, c1_, r1_;\n"
                 "    asm volatile(\"s_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(c0_), \"=s\"(r0_) :: \"memory\");\n") + s[body0:]
end = s.index("    __builtin_amdgcn_s_setprio(0);\n}\n", a)
s = s[:end] + ("    asm volatile(\"s_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(c1_), \"=s\"(r1_) :: \"memory\");\n"
               "    if (SCALE && threadIdx.x == 0 && ke - kb >= 256 && blockIdx.x < 1024 && blockIdx.y == 0 && blockIdx.z == 0) {\n"
               "        g_clk[4 * blockIdx.x + 0] = c1_ - c0_; g_clk[4 * blockIdx.x + 1] = r1_ - r0_; g_clk[4 * blockIdx.x + 2] = ke - kb;\n    }\n") + s[end:]
s = s.replace("hipError_t launch_gemm_grouped(", """void dbg_print_clock() {
    std::vector<unsigned long long> h(4 * 1024, 0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_clk), h.size() * sizeof(unsigned long long));
    std::vector<double> ghz, cpk;
    for (int b = 0; b < 1024; ++b)
        if (h[4 * b + 1] > 0 && h[4 * b + 2] > 0) {
            ghz.push_back((double)h[4 * b] / (double)h[4 * b + 1] * 0.1);
            cpk.push_back((double)h[4 * b] / (double)h[4 * b + 2]);
        }
    if (ghz.empty()) { fprintf(stderr, "no stamps\\n"); return; }
    if (getenv("LPIPM_DIAG_PER_WG"))   // b = blockIdx.x, logical index g = xcd*64 + b/8 (512 workgroups)
        for (int b = 0; b < 512; ++b) fprintf(stderr, "wg %d g %d cpk %.0f\\n", b, (b & 7) * 64 + (b >> 3), (double)h[4 * b] / (double)h[4 * b + 2]);
    std::sort(ghz.begin(), ghz.end()); std::sort(cpk.begin(), cpk.end());
    const size_t n = ghz.size();
    fprintf(stderr, "A.D.A^T data-parallel tiles, %zu workgroups: in-kernel clock median %.3f GHz (min %.3f, max %.3f); "
                    "shader cycles per k-tile (32 MFMAs per wave, 4 waves per SIMD = 8192 MFMA cycles) median %.0f (min %.0f, max %.0f) "
                    "=> MFMA pipe busy %.1f %% of the loop\\n",
            n, ghz[n / 2], ghz[0], ghz[n - 1], cpk[n / 2], cpk[0], cpk[n - 1], 100.0 * 8192.0 / cpk[n / 2]);
}
hipError_t launch_gemm_grouped(""", 1)
open(p, 'w').write(s)
p = 'lp_amd/csrc/solver.hip'
s = open(p).read()
s = s.replace('extern "C" int lpipm_k_adat(', 'namespace lpipm { void dbg_print_clock(); }\nextern "C" int lpipm_k_adat(', 1)
old = "    LP_TRY(timed_repeats(c, repeats, ms_out, [&]() -> int { LP_HIP(run_adat(c, Batch{})); return LPIPM_OK; }));"
assert old in s
s = s.replace(old, old + "\n    lpipm::dbg_print_clock();")
open(p, 'w').write(s)
print("patched: diagnostic build only")
