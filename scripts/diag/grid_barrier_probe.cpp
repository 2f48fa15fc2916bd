// What does a grid-wide barrier cost inside one kernel on this chip?  256 or 512 co-resident workgroups (one or two per CU),
// a monotone counter in device memory: every workgroup's thread 0 adds 1 (agent scope, release), then spins (acquire loads)
// until the count reaches step * nwg.  Reports microseconds per barrier.  This is the hand-off a fused single-kernel
// triangular solve would pay ~16 times per solve.  build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void barriers(unsigned* cnt, int steps, int nwg, double* sink) {
    double acc = 0.0;
    for (int s = 1; s <= steps; ++s) {
        acc += (double)s * 1e-9;                 // (a trace of work between barriers)
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)s * (unsigned)nwg;
            long long spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 50000000LL) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = acc;
}
int main() {
    unsigned* cnt; double* sink;
    hipMalloc(&cnt, 4); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nwg : {64, 256, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
            const int steps = 200;
            hipMemset(cnt, 0, 4);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(barriers, dim3(nwg), dim3(256), 0, 0, cnt, steps, nwg, sink);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%d workgroups: %.2f us per grid barrier (%d barriers in %.1f us)\n", nwg, ms * 1e3 / steps, steps, ms * 1e3);
        }
    }
    return 0;
}
