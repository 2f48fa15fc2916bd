"""Scratch-copy instrumentation of potrf_diag_kernel: where do the ~2900 cycles of the 'priority update' phase (8 tiles of
16x16, one per wave, between two workgroup barriers) go?  Stamps of wave 0 for block column 0."""
p = 'lp_amd/csrc/kernels_potrf.hip'
s = open(p).read()
s = s.replace("__device__ __forceinline__ void update_tiles(double (*Ls)[LS], const double (*xidb)[XS], int jb, bool priority,\n                                             int w, int nw, int fr, int fq) {",
              "__device__ __forceinline__ void update_tiles(double (*Ls)[LS], const double (*xidb)[XS], int jb, bool priority,\n                                             int w, int nw, int fr, int fq, long long* stamps = nullptr) {\n#define ST2(i) do { if (stamps && threadIdx.x == 0) { long long t_; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_) :: \"memory\"); stamps[i] = t_; } } while (0)")
s = s.replace("    for (int t0 = w; t0 < count; t0 += 2 * nw) {\n        const int t1 = t0 + nw;", "    ST2(8);\n    for (int t0 = w; t0 < count; t0 += 2 * nw) {\n        const int t1 = t0 + nw;")
s = s.replace("#pragma unroll\n        for (int u = 0; u < 4; ++u) {\n            ca = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[u], fb0[u], ca, 0, 0, 0);", "        ST2(9);\n#pragma unroll\n        for (int u = 0; u < 4; ++u) {\n            ca = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[u], fb0[u], ca, 0, 0, 0);")
s = s.replace("#pragma unroll\n        for (int r = 0; r < 4; ++r) Ls[r0_ + fq + 4 * r][q0_ + fr] = ca[r];\n        if (has1) {", "        ST2(10);\n#pragma unroll\n        for (int r = 0; r < 4; ++r) Ls[r0_ + fq + 4 * r][q0_ + fr] = ca[r];\n        ST2(11);\n        if (has1) {")
s = s.replace("        update_tiles(Ls, xid[jb & 1], jb, true, wave, NW, fr, fq);          // the next block column first", "        update_tiles(Ls, xid[jb & 1], jb, true, wave, NW, fr, fq, jb == 0 ? stamps : nullptr);          // the next block column first")
open(p, 'w').write(s)
p = 'lp_amd/csrc/solver.hip'
s = open(p).read()
s = s.replace('fprintf(stderr, "diag stamps (cycles): elim(0) %lld, priority update(0) %lld, elim(1)||rest(0) %lld, whole factorisation %lld, write inverses %lld\\n",\n                    h[1]-h[0], h[2]-h[1], h[3]-h[2], h[6]-h[0], h[7]-h[6]);',
              'fprintf(stderr, "diag stamps (cycles): elim(0) %lld, priority update(0) %lld [enter %lld, loads done %lld, mfma done %lld, stores done %lld, barrier %lld], elim(1)||rest(0) %lld, whole %lld\\n",\n                    h[1]-h[0], h[2]-h[1], h[8]-h[1], h[9]-h[8], h[10]-h[9], h[11]-h[10], h[2]-h[11], h[3]-h[2], h[6]-h[0]);')
open(p, 'w').write(s)
print("patched")
