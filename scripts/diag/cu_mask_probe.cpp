// Which CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  Prints, per mask pattern, the set of
// (XCC_ID, HW_ID) pairs observed by 2048 one-wave workgroups.  Build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <set>
__global__ void where(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // spin a little so that the grid spreads over every eligible CU
    unsigned long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int nb = 4096;
    unsigned* d; hipMalloc(&d, nb * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;   // xcc -> set of (se, cu)
    for (int i = 0; i < nb; ++i) {
        const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    size_t total = 0;
    printf("%s:", name);
    for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  -> %zu distinct CUs\n", total);
    hipFree(d); hipStreamDestroy(st);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("all 256 bits", all);
    std::vector<uint32_t> first32(8, 0); first32[0] = 0xffffffffu;
    run("bits 0-31", first32);
    std::vector<uint32_t> every8(8, 0x01010101u);
    run("every 8th bit (32 bits)", every8);
    std::vector<uint32_t> low4of32(8, 0x0000000fu);
    run("bits 0-3 of every word (32 bits)", low4of32);
    std::vector<uint32_t> not_low4(8, 0xfffffff0u);
    run("all but bits 0-3 of every word (224 bits)", not_low4);
    std::vector<uint32_t> one(8, 0); one[0] = 1;
    run("bit 0 only", one);
    std::vector<uint32_t> w1(1, 0x0000000fu);
    run("mask of ONE word, bits 0-3", w1);
    std::vector<uint32_t> w1b(1, 0xfffffff0u);
    run("mask of ONE word, all but bits 0-3", w1b);
    return 0;
}
