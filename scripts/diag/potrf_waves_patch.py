"""Scratch-copy instrumentation of potrf_diag_kernel: for block column 0, every wave's SIMD id (HW_ID) and the cycles it
spends in its part of the overlapped phase (waves 0..2: elimination of column 1; waves 3..: rest of update 0)."""
p = 'lp_amd/csrc/kernels_potrf.hip'
s = open(p).read()
s = s.replace("""        if (wave < 3) {
            __builtin_amdgcn_s_setprio(3);       // the serial chain outranks the throughput work sharing its SIMDs""", """        long long tw0_ = 0, tw1_ = 0;
        if (stamps && jb == 0) asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(tw0_) :: "memory");
        if (wave < 3) {
            __builtin_amdgcn_s_setprio(3);       // the serial chain outranks the throughput work sharing its SIMDs""")
s = s.replace("""            store_block_column(jb, tid - 192, DT - 192);
        }""", """            if (stamps && jb == 0) asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(tw1_) :: "memory");
            store_block_column(jb, tid - 192, DT - 192);
        }
        if (stamps && jb == 0) {
            if (wave < 3) asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(tw1_) :: "memory");
            if (lane == 0) { stamps[32 + 2 * wave] = tw1_ - tw0_; stamps[33 + 2 * wave] = (__builtin_amdgcn_s_getreg(0x1904) ); }
        }""")
open(p, 'w').write(s)
p = 'lp_amd/csrc/solver.hip'
s = open(p).read()
s = s.replace("long long* d = nullptr; long long h[16] = {0};", "long long* d = nullptr; long long h[80] = {0};")
s = s.replace('fprintf(stderr, "diag stamps (cycles): elim(0) %lld, priority update(0) %lld, elim(1)||rest(0) %lld, whole factorisation %lld, write inverses %lld\\n",\n                    h[1]-h[0], h[2]-h[1], h[3]-h[2], h[6]-h[0], h[7]-h[6]);',
              'fprintf(stderr, "overlapped phase of block column 0: elim(1)||rest(0) %lld cycles; per wave (simd: cycles):", h[3]-h[2]);\n            for (int w = 0; w < 16; ++w) fprintf(stderr, " w%d(s%lld: %lld)", w, h[33+2*w] & 3, h[32+2*w]);\n            fprintf(stderr, "\\n");')
open(p, 'w').write(s)
print("patched")
