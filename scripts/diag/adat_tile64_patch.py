"""Scratch-copy experiment (timing only, results unscaled): A.D.A^T as 64x64 output tiles, one per workgroup, 4 workgroups
(4 waves per SIMD) per CU, through gemm_nt_tile_kernel<2,2>."""
p = 'lp_amd/csrc/solver.hip'
s = open(p).read()
s = s.replace("    g.diag_pad_from = (int)c->m; g.ws = c->ws; g.nwg = c->adat_nwg; g.batch = bt;",
              "    g.diag_pad_from = -1; g.ws = nullptr; g.s = nullptr; g.tile_edge = 64; g.tile_list = nullptr;\n    { const int t64 = c->mp / 64; g.ntiles = t64 * (t64 + 1) / 2; } g.nwg = g.ntiles; g.batch = bt;")
open(p, 'w').write(s)
print("patched: 64x64 tiles")
