// Instruction-latency probe for the pivot chain of potrf_diag_kernel (gfx950): one wave, shader-clock cycles per
// dependent operation.  Build: hipcc --offload-arch=gfx950 -O2 -o latency_probe.bin latency_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

__global__ void probe(double* out, unsigned long long* cyc, double seed) {
    __shared__ double lds[64];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3, y = 1.0000001, z = 0.5;
    unsigned long long t0, t1;
    int k = 0;
    // 0: empty
    t0 = now(); t1 = now(); cyc[k++] = t1 - t0;
    // 1: dependent v_fma_f64
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    t1 = now(); cyc[k++] = t1 - t0;
    // 2: independent v_fma_f64 (4 accumulators)
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(y), "v"(z));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(y), "v"(z));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(y), "v"(z));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(y), "v"(z));
    }
    t1 = now(); cyc[k++] = t1 - t0;
    x += a0 + a1 + a2 + a3;
    // 3: dependent v_rcp_f64
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
    t1 = now(); cyc[k++] = t1 - t0;
    // 4: dependent v_mul_f64
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    t1 = now(); cyc[k++] = t1 - t0;
    // 5: fma -> readlane(lo,hi) -> fma reading the SGPR pair
    {
        double s;
        t0 = now();
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            s = readlane_f64(x, 3);
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x) : "s"(s), "v"(y));
        }
        t1 = now(); cyc[k++] = t1 - t0;
    }
    // 6: the pivot chain as shipped: rcp(s) + 2 Newton steps + mul + fma + readlane
    {
        double piv = readlane_f64(x, 0);
        double a = x, c1 = 0.25;
        t0 = now();
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            double r, e;
            asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "s"(piv));
            asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "s"(piv), "v"(r));
            asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r) : "v"(e));
            asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "s"(piv), "v"(r));
            asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r) : "v"(e));
            asm volatile("v_mul_f64 %0, %1, -%0" : "+v"(r) : "v"(a));
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(c1), "v"(r));
            piv = readlane_f64(a, 5);
        }
        t1 = now(); cyc[k++] = t1 - t0;
        x += a;
    }
    // 7: the shortened chain: rcp, e, s = e + e*e, a' = fma(-(t0*c1), s, w0), readlane (t0, w0, x off the chain)
    {
        double piv = readlane_f64(x, 0);
        double a = x, c1 = 0.25;
        t0 = now();
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            double r, e, s, t0_, w0, xx;
            asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "s"(piv));
            asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "s"(piv), "v"(r));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t0_) : "v"(a), "v"(r));
            asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(s) : "v"(e));
            asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(w0) : "v"(t0_), "v"(c1), "v"(a));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(xx) : "v"(t0_), "v"(c1));
            asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(a) : "v"(xx), "v"(s), "v"(w0));
            piv = readlane_f64(a, 5);
        }
        t1 = now(); cyc[k++] = t1 - t0;
        x += a;
    }
    // 8: LDS round trip: ds_write_b64 -> wave barrier -> ds_read_b64 (dependent)
    {
        t0 = now();
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            lds[lane] = x;
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x = lds[(lane + 1) & 63];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        t1 = now(); cyc[k++] = t1 - t0;
    }
    // 9: dependent v_rsq_f64
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
    t1 = now(); cyc[k++] = t1 - t0;
    // 10: dependent MFMA 16x16x4 f64 on one accumulator
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 c = {x, x, x, x};
        t0 = now();
#pragma unroll
        for (int i = 0; i < REP; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c, 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        t1 = now(); cyc[k++] = t1 - t0;
        x += c[0] + c[1] + c[2] + c[3];
    }
    // 11: accuracy of the seeds: relative error of v_rcp_f64 / v_rsq_f64 on this lane's value
    {
        const double d = 1.0 + lane * 0.0153 + seed * 1e-3;
        double r0, q0;
        asm volatile("v_rcp_f64 %0, %1" : "=v"(r0) : "v"(d));
        asm volatile("v_rsq_f64 %0, %1" : "=v"(q0) : "v"(d));
        out[64 + lane] = fma(-d, r0, 1.0);
        out[128 + lane] = fma(-d * q0, q0, 1.0);
        const double e = fma(-d, r0, 1.0);
        const double s = fma(e, e, e);
        const double r3 = fma(r0, s, r0);               // three-term series
        double r2 = r0;
        for (int it = 0; it < 2; ++it) r2 = fma(r2, fma(-d, r2, 1.0), r2);
        out[192 + lane] = (r3 - r2) / r2;
    }
    out[lane] = x;
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * sizeof(double)); hipMalloc(&cyc, 16 * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.2345);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16); std::vector<double> o(256);
    hipMemcpy(h.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"empty", "dep v_fma_f64", "indep v_fma_f64", "dep v_rcp_f64", "dep v_mul_f64", "readlane x2 -> fma(sgpr)",
                           "pivot chain (shipped)", "pivot chain (short)", "lds write->read", "dep v_rsq_f64", "dep mfma 16x16x4"};
    for (int i = 0; i < 11; ++i)
        printf("%-28s %6llu cycles total, %.1f per op (minus empty)\n", names[i], h[i], ((double)h[i] - (double)h[0]) / REP);
    double me = 0, mq = 0, md = 0;
    for (int l = 0; l < 64; ++l) { me = fmax(me, fabs(o[64 + l])); mq = fmax(mq, fabs(o[128 + l])); md = fmax(md, fabs(o[192 + l])); }
    printf("seed error: rcp %.2e  rsq(1 - d q^2) %.2e;  three-term vs two Newton steps: %.2e\n", me, mq, md);
    return 0;
}
