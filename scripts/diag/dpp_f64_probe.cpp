// v_fmac_f64_dpp / v_rcp_f64_dpp / v_mov_b64_dpp with row_newbcast on gfx950: issue rate, dependent latency, and a
// check of the broadcast semantics.  Build: hipcc --offload-arch=gfx950 -O2 -o dpp_f64_probe.bin dpp_f64_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__global__ void probe(double* out, unsigned long long* cyc) {
    const int lane = threadIdx.x;
    double x = 1.0 + lane * 1e-3, y = 1.0000001, z = 0.5;
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    unsigned long long t0, t1;
    int k = 0;
    t0 = now(); t1 = now(); cyc[k++] = t1 - t0;
    // 1: independent v_fmac_f64_dpp (4 accumulators)
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(y), "v"(z));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(y), "v"(z));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(y), "v"(z));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(y), "v"(z));
    }
    t1 = now(); cyc[k++] = t1 - t0;
    // 2: dependent through the accumulator
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(y), "v"(z));
    t1 = now(); cyc[k++] = t1 - t0;
    // 3: dependent through the DPP source (written by the previous op): needs the wait states
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP / 2; ++i) {
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(a0), "v"(z));
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(a1), "v"(z));
    }
    t1 = now(); cyc[k++] = t1 - t0;
    // 4: dependent v_rcp_f64_dpp
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("s_nop 1\n\tv_rcp_f64_dpp %0, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(x));
    t1 = now(); cyc[k++] = t1 - t0;
    // 5: independent plain v_fmac_f64 for reference
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(y), "v"(z));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a1) : "v"(y), "v"(z));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a2) : "v"(y), "v"(z));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a3) : "v"(y), "v"(z));
    }
    t1 = now(); cyc[k++] = t1 - t0;
    // 6: plain fmac -> DPP consumer of its result with s_nop 1 (the chain link of the pivot loop)
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP / 2; ++i) {
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(z));
        asm volatile("s_nop 1\n\tv_rcp_f64_dpp %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "=v"(a1) : "v"(a0));
    }
    t1 = now(); cyc[k++] = t1 - t0;
    // semantics: dst += bcast_k(src0) * src1 per row of 16 lanes
    {
        double s0 = 100.0 * (lane >> 4) + (lane & 15), s1 = 2.0, d = 0.25;
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s0), "v"(s1));
        out[64 + lane] = d;        // expect 0.25 + 2 * (100 * row + 5)
        double m;
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(s0));
        out[128 + lane] = m;       // expect 100 * row + 7
    }
    out[lane] = x + a0 + a1 + a2 + a3;
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * sizeof(double)); hipMalloc(&cyc, 16 * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16); std::vector<double> o(256);
    hipMemcpy(h.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"empty", "indep v_fmac_f64_dpp", "dep (accumulator) v_fmac_f64_dpp", "dep (dpp source, s_nop 1)", "dep v_rcp_f64_dpp (s_nop 1)",
                           "indep v_fmac_f64", "fmac -> s_nop 1 -> rcp_dpp (per pair)"};
    for (int i = 0; i < 7; ++i) printf("%-40s %6llu total, %.1f per op\n", names[i], h[i], ((double)h[i] - (double)h[0]) / (i == 6 ? REP / 2 : REP));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (o[64 + l] != 0.25 + 2.0 * (100.0 * (l >> 4) + 5)) ++bad;
        if (o[128 + l] != 100.0 * (l >> 4) + 7) ++bad;
    }
    printf("broadcast semantics: %s (lane 0: %g %g, lane 37: %g %g)\n", bad ? "MISMATCH" : "ok", o[64], o[128], o[64 + 37], o[128 + 37]);
    return 0;
}
