#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int K> __device__ __forceinline__ void fmac_bcast(double& acc, const double& src, const double& m) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(K));
}
template <int K> __device__ __forceinline__ double rcp_bcast(const double& src) {
    double r;
    asm volatile("s_nop 1\n\tv_rcp_f64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(K));
    return r;
}
template <int K> __device__ __forceinline__ double mov_bcast(const double& src) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(K));
    return r;
}
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    double a0 = 500.0 + lane, a1 = 3.0 + 0.1 * lane;
    double rd = rcp_bcast<0>(a0);     // reads its source as zero: inf
    const double pv = mov_bcast<0>(a0);
    double r = __builtin_amdgcn_rcp(pv);
    out[lane] = rd; out[64 + lane] = pv;
    double e = fma(-pv, r, 1.0);
    r = fma(r, e, r);
    e = fma(-pv, r, 1.0);
    r = fma(r, e, r);
    out[128 + lane] = r;
    double tna = -a0 * r;
    fmac_bcast<1>(a1, a0, tna);
    out[192 + lane] = a1;
}
int main() {
    double* out; hipMalloc(&out, 256 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out);
    hipDeviceSynchronize();
    std::vector<double> o(256); hipMemcpy(o.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 17, 35}) printf("lane %d: v_rcp_f64_dpp %.6g (1/pv = %.6g)  pv %.6g  r %.6g  a1 %.6g (exp %.6g)\n", l, o[l], 1.0 / (500.0 + (l & ~15)), o[64 + l], o[128 + l],
                                           o[192 + l], 3.0 + 0.1 * l - (500.0 + l) / (500.0 + (l & ~15)) * (500.0 + (l & ~15) + 1));
    return 0;
}
