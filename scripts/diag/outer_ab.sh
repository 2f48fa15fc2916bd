# A/B of the outer-panel width of the factorisation (POTRF_OUTER, compile-time): builds two copies of the library on the box
set -e
R=$GRAFT_REPO_ROOT
for W in 2 3; do
  rm -rf /tmp/lp_$W && mkdir -p /tmp/lp_$W && cp -r $R/lp_amd $R/include $R/scripts $R/bench.py $R/oracle $R/profiles /tmp/lp_$W/ 2>/dev/null
  sed -i "s/constexpr int POTRF_OUTER = 4;/constexpr int POTRF_OUTER = $W;/" /tmp/lp_$W/lp_amd/csrc/lpipm_internal.hpp
  (cd /tmp/lp_$W/lp_amd/csrc && make clean >/dev/null && make -j8 >/dev/null 2>&1)
  echo "OUTER=$W"; (cd /tmp/lp_$W && python scripts/potrf_repeat.py 4096 8 | tail -1 && python bench.py --only-headline --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['phase_ms_per_iteration'])")
done
