cd /tmp && export TMPDIR=/tmp
export LPIPM_EXPERIMENTAL=1 LPIPM_HALVES=0
R=$GRAFT_REPO_ROOT
for K in 32 8; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr$K -o t -- python3 $R/scripts/lockstep_c4.py $K 1024 2048 2 > $R/gpurun_out/tr$K.log 2>&1
  f=$(find $R/gpurun_out/tr$K -name '*kernel_trace.csv' | head -1)
  python3 $R/scripts/trace_iter.py $f -3 > $R/gpurun_out/trace_c4_$K.txt
  rm -rf $R/gpurun_out/tr$K
done
tail -3 $R/gpurun_out/tr32.log
