// FP64 MFMA (v_mfma_f64_16x16x4_f64) on gfx950: latency of a dependent chain, issue rate of independent chains in one
// wave, and what waves on the same / on different SIMDs get when they run chains together.  One workgroup of 16 waves.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_f64_probe.bin mfma_f64_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 32
__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
// mode 0: one dependent chain per active wave; mode 1: four independent chains per active wave; mode 2: dependent
// v_fma_f64 chain beside (for the waves with bit set in fma_mask) -- active waves: bit w of mask
__global__ void probe(double* out, unsigned long long* cyc, unsigned mask, unsigned fma_mask, int mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int simd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(simd));
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
    d4 c0 = {a, a, a, a}, c1 = c0, c2 = c0, c3 = c0;
    double x = a;
    __syncthreads();
    unsigned long long t0 = now();
    if ((mask >> wave) & 1) {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < REP; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int i = 0; i < REP / 4; ++i) {
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("" : "+v"(c0) :: "memory");
        asm volatile("" : "+v"(c1) :: "memory");
        asm volatile("" : "+v"(c2) :: "memory");
        asm volatile("" : "+v"(c3) :: "memory");
    } else if ((fma_mask >> wave) & 1) {
#pragma unroll
        for (int i = 0; i < 8 * REP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
    }
    unsigned long long t1 = now();
    if ((threadIdx.x & 63) == 0) { cyc[2 * wave] = t1 - t0; cyc[2 * wave + 1] = simd; }
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x;
}
static void run(const char* name, unsigned mask, unsigned fma_mask, int mode) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * sizeof(double)); hipMalloc(&cyc, 32 * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(1024), 0, 0, out, cyc, mask, fma_mask, mode);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(32);
    hipMemcpy(h.data(), cyc, 32 * 8, hipMemcpyDeviceToHost);
    printf("%-58s", name);
    for (int w = 0; w < 16; ++w)
        if (((mask | fma_mask) >> w) & 1) printf(" w%d(s%llu):%.1f", w, h[2 * w + 1], (double)h[2 * w] / (((mask >> w) & 1) ? REP : 8 * REP));
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run("1 wave, dependent chain (cycles per MFMA)", 0x1, 0, 0);
    run("1 wave, 4 independent chains", 0x1, 0, 1);
    run("4 waves on 4 SIMDs (w0-3), dependent", 0xF, 0, 0);
    run("4 waves on 4 SIMDs (w0-3), independent", 0xF, 0, 1);
    run("4 waves on ONE SIMD (w0,4,8,12), dependent", 0x1111, 0, 0);
    run("4 waves on ONE SIMD (w0,4,8,12), independent", 0x1111, 0, 1);
    run("2 waves on one SIMD (w0,4), dependent", 0x11, 0, 0);
    run("16 waves, dependent", 0xFFFF, 0, 0);
    run("16 waves, independent", 0xFFFF, 0, 1);
    run("12 waves on 3 SIMDs (not w%4==0), dependent", 0xEEEE, 0, 0);
    run("w1 MFMA dependent beside w0 v_fma_f64 chain (other SIMD)", 0x2, 0x1, 0);
    run("w4 MFMA dependent beside w0 v_fma_f64 chain (same SIMD)", 0x10, 0x1, 0);
    run("w0 v_fma_f64 chain alone (cycles per fma)", 0x0, 0x1, 0);
    run("w0 v_fma_f64 chain beside 12 MFMA waves on other SIMDs", 0xEEEE, 0x1, 0);
    return 0;
}
