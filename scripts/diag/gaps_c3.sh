cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_c3 -o t -- python3 $R/scripts/prof_c3.py > /tmp/tr_c3.log 2>&1
python3 $R/scripts/trace_gaps_all.py $(find /tmp/tr_c3 -name '*kernel_trace.csv' | head -1) 10
