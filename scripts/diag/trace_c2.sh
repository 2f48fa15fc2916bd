cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trc2 -o t -- python3 $R/scripts/prof_c3.py 512 1024 > $R/gpurun_out/trc2.log 2>&1
f=$(find $R/gpurun_out/trc2 -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/trace_iter.py $f -3 k_fused_residuals > $R/gpurun_out/trace_c2.txt
rm -rf $R/gpurun_out/trc2
