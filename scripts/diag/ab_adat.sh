#!/bin/bash
# on the GPU box, in the scratch snapshot: build each main-loop variant with the clock stamps and run it.
# The variants of adat_variant_patch.py edit the 4-wave main loop (tile_mainloop), which A.D.A^T uses only with
# LPIPM_ADAT_W4=1; the numbers in profiles/r01_adat_loop_diag.txt were taken when that kernel was the default, with a
# version of adat_clock_patch.py that stamped it (git history).  The shipped 8-wave loop is stamped by the current one.
set -e
mkdir -p gpurun_out/dbg
cp lp_amd/csrc/kernels_gemm.hip /tmp/kernels_gemm.orig; cp lp_amd/csrc/solver.hip /tmp/solver.orig
for v in "$@"; do
  cp /tmp/kernels_gemm.orig lp_amd/csrc/kernels_gemm.hip; cp /tmp/solver.orig lp_amd/csrc/solver.hip
  python scripts/diag/adat_variant_patch.py $v
  python scripts/diag/adat_clock_patch.py > /dev/null
  make -C lp_amd/csrc -j16 > gpurun_out/dbg/make_$v.log 2>&1 || { tail -20 gpurun_out/dbg/make_$v.log; exit 1; }
  echo "== $v"
  timeout -k 10 200 python scripts/diag/adat_clock_run.py 2>&1 | grep -v amdgpu.ids | tail -2
done
