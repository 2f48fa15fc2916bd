// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  Two spin kernels of ~100 us on 16 workgroups each:
// back to back (plain launches) vs the second one launched with the any-order flag.  build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    int* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long cyc = 10000;   // wall_clock64 ticks at 100 MHz: 100 us
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, st, cyc, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, st, cyc, d);
            else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(16), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
            else { hipExtLaunchKernelGGL(spin, dim3(16), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
                   hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, st, cyc, d); }
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s): %.1f us\n", mode, mode == 0 ? "plain, plain" : mode == 1 ? "plain, any-order" : "plain, any-order, plain", ms * 1e3);
        }
    }
    return 0;
}
