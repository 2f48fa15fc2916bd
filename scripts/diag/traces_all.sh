# The kernel traces quoted in DESIGN.md, from the build as it is: one C3 iteration, one C2 iteration, one C4 lockstep
# iteration (one stream), the two half-batch views side by side, one factorisation at m = 4096 -> gpurun_out/traces/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/traces
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_c3 -o t -- python3 $R/scripts/prof_c3.py > /tmp/tr_c3.log 2>&1
python3 $R/scripts/trace_iter.py $(find /tmp/tr_c3 -name '*kernel_trace.csv' | head -1) -3 > $R/gpurun_out/traces/r03_c3_iteration_trace.txt
bash $R/scripts/diag/trace_c2.sh; cp $R/gpurun_out/trace_c2.txt $R/gpurun_out/traces/r03_c2_iteration_trace.txt
bash $R/scripts/diag/trace_c4.sh > /dev/null 2>&1; cp $R/gpurun_out/trace_c4_32.txt $R/gpurun_out/traces/r03_c4_iteration_trace.txt; cp $R/gpurun_out/trace_c4_8.txt $R/gpurun_out/traces/r03_c4_iteration_trace_batch8.txt
bash $R/scripts/diag/trace_c4_halves.sh > /dev/null 2>&1; cp $R/gpurun_out/trace_c4_halves.txt $R/gpurun_out/traces/r03_c4_halves_trace.txt
bash $R/scripts/diag/trace_potrf4096.sh > /dev/null 2>&1; cp $R/gpurun_out/trace_potrf4096.txt $R/gpurun_out/traces/r03_potrf4096_trace.txt
ls -la $R/gpurun_out/traces
