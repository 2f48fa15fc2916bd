// kernels_probe.hip -- fp64 MFMA issue-rate probe: confirms the roofline denominator on the box.
// 4 waves per workgroup, 16 independent accumulators per wave, operands in registers, no memory
// traffic in the loop: the achieved rate is what v_mfma_f64_16x16x4_f64 can issue, chip-wide.
#include "lpipm_internal.hpp"

namespace lpipm {

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void mfma_f64_probe_kernel(int iters, double* sink) {
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * (double)threadIdx.x, b = 1.0 - 1e-9 * (double)threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a += 1e-12;  // keeps the loop from being collapsed; random-ish, non-zero operands
        b -= 1e-12;
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

hipError_t launch_mfma_probe(int iters, double* sink, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, st, iters, sink);
    return hipGetLastError();
}

}  // namespace lpipm
