cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trh -o t -- python3 $R/scripts/lockstep_c4.py 32 1024 2048 3 > $R/gpurun_out/trh.log 2>&1
f=$(find $R/gpurun_out/trh -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/trace_window.py $f 0 3500 > $R/gpurun_out/trace_c4_halves.txt
rm -rf $R/gpurun_out/trh
grep lockstep $R/gpurun_out/trh.log
