cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trp -o t -- python3 $R/scripts/potrf_repeat.py 4096 4 > $R/gpurun_out/trp.log 2>&1
f=$(find $R/gpurun_out/trp -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/trace_factor.py $f > $R/gpurun_out/trace_potrf4096.txt
rm -rf $R/gpurun_out/trp
grep "m=" $R/gpurun_out/trp.log
