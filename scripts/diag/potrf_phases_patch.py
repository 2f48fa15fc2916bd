"""Scratch-copy instrumentation of potrf_diag_kernel: cycles of every phase of every block column (wave 0):
priority update + barrier, then elimination||rest + barrier."""
p = 'lp_amd/csrc/kernels_potrf.hip'
s = open(p).read()
s = s.replace("        lds_barrier();\n        if (jb == 0) STAMP(2);", "        lds_barrier();\n        STAMP(8 + 2 * jb);")
s = s.replace("        lds_barrier();\n        if (jb == 0) STAMP(3);", "        lds_barrier();\n        STAMP(9 + 2 * jb);")
open(p, 'w').write(s)
p = 'lp_amd/csrc/solver.hip'
s = open(p).read()
s = s.replace("long long* d = nullptr; long long h[16] = {0};", "long long* d = nullptr; long long h[32] = {0};")
s = s.replace('fprintf(stderr, "diag stamps (cycles): elim(0) %lld, priority update(0) %lld, elim(1)||rest(0) %lld, whole factorisation %lld, write inverses %lld\\n",\n                    h[1]-h[0], h[2]-h[1], h[3]-h[2], h[6]-h[0], h[7]-h[6]);',
              'fprintf(stderr, "diag phases (cycles): load->elim(0) %lld |", h[1]-h[0]);\n            { long long prev = h[1]; for (int jb = 0; jb < 7; ++jb) { fprintf(stderr, " jb%d prio %lld elim||rest %lld |", jb, h[8+2*jb]-prev, h[9+2*jb]-h[8+2*jb]); prev = h[9+2*jb]; } }\n            fprintf(stderr, " whole %lld, final stores %lld\\n", h[6]-h[0], h[7]-h[6]);')
open(p, 'w').write(s)
print("patched")
