#!/bin/bash
# Profiles of one round, on the GPU box:  bash scripts/collect_profiles.sh r03
# -> gpurun_out/<round>_profiles/: rocprofv3 kernel stats of the default bench command and of the lockstep C4 shard, PMC
#    passes (one counter group per pass: FETCH_SIZE and WRITE_SIZE do not fit one pass) summarised by scripts/pmc_summary.py
#    with the hash of lp_amd/csrc, and the bench JSON lines (default = c3 with baselines + c2 + c4 objects; c2; c4).
#    Copy what is to be judged into profiles/.
set -o pipefail
R=${1:-r03}
OUT=$PWD/gpurun_out/${R}_profiles
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o bench --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --only-headline > $OUT/${R}_bench_profiled.json 2> /tmp/p_bench.log
cp /tmp/p_bench/bench_kernel_stats.csv $OUT/${R}_bench_kernel_stats.csv
echo "kernel stats of the headline done"
LPIPM_EXPERIMENTAL=1 LPIPM_HALVES=0 rocprofv3 --kernel-trace --stats -d /tmp/p_c4 -o c4 --output-format csv -- python3 $REPO/scripts/lockstep_c4.py 32 1024 2048 5 > $OUT/${R}_c4_lockstep_profiled.txt 2> /tmp/p_c4.log
cp /tmp/p_c4/c4_kernel_stats.csv $OUT/${R}_c4_lockstep_kernel_stats.csv
echo "kernel stats of the lockstep shard done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  tag=$(echo $grp | cut -d" " -f1)
  rocprofv3 --pmc $grp -d /tmp/p_pmc_c3/$tag -o pmc --output-format csv -- python3 $REPO/scripts/prof_c3.py > /tmp/p_pmc_$tag.log 2>&1
  rocprofv3 --pmc $grp -d /tmp/p_pmc_c2/$tag -o pmc --output-format csv -- python3 $REPO/scripts/prof_c3.py 512 1024 > /tmp/p_pmc2_$tag.log 2>&1
  LPIPM_EXPERIMENTAL=1 LPIPM_HALVES=0 rocprofv3 --pmc $grp -d /tmp/p_pmc_c4/$tag -o pmc --output-format csv -- python3 $REPO/scripts/lockstep_c4.py 32 1024 2048 2 > /tmp/p_pmc4_$tag.log 2>&1
  echo "pmc pass $tag done"
done
cd $REPO
PMC_LPS_PER_LAUNCH=1 python3 scripts/pmc_summary.py gemm_nt_units 4096 8192 $OUT/${R}_adat_pmc.json /tmp/p_pmc_c3 > /dev/null
python3 scripts/pmc_summary.py gemv_dual 512 1024 $OUT/${R}_gemv_pmc.json /tmp/p_pmc_c2 > /dev/null
python3 scripts/pmc_summary.py gemm_nt_units 1024 2048 $OUT/${R}_adat_c4_pmc.json /tmp/p_pmc_c4 > /dev/null
# the bench line reports the PMC figures only from a summary whose csrc hash matches: put the fresh ones where it looks
cp $OUT/${R}_adat_pmc.json $OUT/${R}_gemv_pmc.json $OUT/${R}_adat_c4_pmc.json profiles/
python3 bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err
python3 bench.py --workload c2 --no-cpu-baseline > $OUT/${R}_bench_c2.json 2>> $OUT/${R}_bench.err
python3 bench.py --workload c4 --steps 5 > $OUT/${R}_bench_c4.json 2>> $OUT/${R}_bench.err
ls -la $OUT
