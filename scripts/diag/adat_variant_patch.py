"""Scratch-copy variants of the A.D.A^T main loop for A/B runs on the box (never committed as such).
usage: adat_variant_patch.py <variant>   then the clock patch, make, adat_clock_run.py"""
import sys
v = sys.argv[1]
p = 'lp_amd/csrc/kernels_gemm.hip'
s = open(p).read()
loop_old = """#pragma unroll
        for (int round = 0; round < 2; ++round) {
            d2 a[MTM], b[MTN];
#pragma unroll
            for (int mi = 0; mi < MTM; ++mi)
                a[mi] = *(const d2*)&ldsA[cur][wr * (16 * MTM) + mi * 16 + fr][round * 8 + fq * 2];
#pragma unroll
            for (int nj = 0; nj < MTN; ++nj)
                b[nj] = *(const d2*)&ldsB[cur][wc * (16 * MTN) + nj * 16 + fr][round * 8 + fq * 2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mi = 0; mi < MTM; ++mi)
#pragma unroll
                    for (int nj = 0; nj < MTN; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
}"""
assert loop_old in s
if v == "prio_overhead":      # the wave that is NOT in its MFMA phase gets issue priority
    new = loop_old.replace("            d2 a[MTM], b[MTN];\n", "            d2 a[MTM], b[MTN];\n            if (round == 0) __builtin_amdgcn_s_setprio(0);\n", 1)
    new = new.replace("        if (more) lstore(cur ^ 1);", "        __builtin_amdgcn_s_setprio(2);\n        if (more) lstore(cur ^ 1);")
elif v == "prio_mfma":        # the wave in its MFMA phase gets priority
    new = loop_old.replace("#pragma unroll\n            for (int t = 0; t < 2; ++t)", "            __builtin_amdgcn_s_setprio(2);\n#pragma unroll\n            for (int t = 0; t < 2; ++t)", 1)
    new = new.replace("        }\n        if (more) lstore(cur ^ 1);", "            __builtin_amdgcn_s_setprio(0);\n        }\n        if (more) lstore(cur ^ 1);")
elif v == "prio_noscale":     # timing only (wrong numbers): no x/z multiply, no load of the scale vector
    new = loop_old.replace("            d2 a[MTM], b[MTN];\n", "            d2 a[MTM], b[MTN];\n            if (round == 0) __builtin_amdgcn_s_setprio(0);\n", 1)
    new = new.replace("        if (more) lstore(cur ^ 1);", "        __builtin_amdgcn_s_setprio(2);\n        if (more) lstore(cur ^ 1);")
    s = s.replace("= SCALE ? sb[r] * sv : sb[r];", "= sb[r];")
    s = s.replace("        if (SCALE) sv = *(const d2*)(s + ko + scol);\n", "")
elif v == "prio_stride20":    # + LDS row stride 20 doubles: ds_read_b128 lane groups conflict-free (18 is 2-way)
    new = loop_old.replace("            d2 a[MTM], b[MTN];\n", "            d2 a[MTM], b[MTN];\n            if (round == 0) __builtin_amdgcn_s_setprio(0);\n", 1)
    new = new.replace("        if (more) lstore(cur ^ 1);", "        __builtin_amdgcn_s_setprio(2);\n        if (more) lstore(cur ^ 1);")
    s = s.replace("constexpr int LDS_STRIDE = BK + 2;", "constexpr int LDS_STRIDE = BK + 4;")
elif v == "stride20":
    new = loop_old
    s = s.replace("constexpr int LDS_STRIDE = BK + 2;", "constexpr int LDS_STRIDE = BK + 4;")
elif v == "prio_ramp":        # overhead prio 3; MFMA phase prio rises with progress (round 0: 1, round 1: 2): the wave further into its
                              # MFMA phase keeps the pipe, a wave just entering it waits -> the two waves of a SIMD fall into anti-phase
    new = loop_old.replace("#pragma unroll\n            for (int t = 0; t < 2; ++t)", "            if (round == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);\n#pragma unroll\n            for (int t = 0; t < 2; ++t)", 1)
    new = new.replace("        if (more) lstore(cur ^ 1);", "        __builtin_amdgcn_s_setprio(3);\n        if (more) lstore(cur ^ 1);")
elif v == "prio_ramp4":       # same with four levels inside the MFMA phase (per t), overhead prio 3 as well
    new = loop_old.replace("#pragma unroll\n            for (int t = 0; t < 2; ++t)\n#pragma unroll\n                for (int mi = 0; mi < MTM; ++mi)", "#pragma unroll\n            for (int t = 0; t < 2; ++t) {\n                if (round * 2 + t == 0) __builtin_amdgcn_s_setprio(0); else if (round * 2 + t == 1) __builtin_amdgcn_s_setprio(1); else if (round * 2 + t == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);\n#pragma unroll\n                for (int mi = 0; mi < MTM; ++mi)", 1)
    new = new.replace("                        acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);\n        }", "                        acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][t], b[nj][t], acc[mi][nj], 0, 0, 0);\n            }\n        }")
    new = new.replace("        if (more) lstore(cur ^ 1);", "        __builtin_amdgcn_s_setprio(3);\n        if (more) lstore(cur ^ 1);")
elif v.startswith("prio_dephase"):   # prio_overhead + the wave in the odd hardware slot of its SIMD starts half an iteration later
    n = v[len("prio_dephase"):] or "65"
    new = loop_old.replace("            d2 a[MTM], b[MTN];\n", "            d2 a[MTM], b[MTN];\n            if (round == 0) __builtin_amdgcn_s_setprio(0);\n", 1)
    new = new.replace("        if (more) lstore(cur ^ 1);", "        __builtin_amdgcn_s_setprio(2);\n        if (more) lstore(cur ^ 1);")
    s = s.replace("    const int KT = p.KT;\n    const int ntiles_dp = (p.ntiles / p.nwg) * p.nwg;\n\n    for (int tile = g;",
                  "    const int KT = p.KT;\n    const int ntiles_dp = (p.ntiles / p.nwg) * p.nwg;\n    if (__builtin_amdgcn_s_getreg(0x1804) & 1) __builtin_amdgcn_s_sleep(" + n + ");   // HW_ID.wave_id\n\n    for (int tile = g;")
    assert "s_sleep" in s
elif v == "base":
    new = loop_old
else:
    raise SystemExit("unknown variant")
s = s.replace(loop_old, new)
open(p, 'w').write(s)
print("variant", v)
