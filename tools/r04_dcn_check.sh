#!/bin/bash
# tools/r04_dcn_check.sh <tag>: DCN op parity + Lore parity + tsr-only kernel stats (round-4 DCN work)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04b}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dcn_op.py tests/test_gpu_tsr.py -x -q -s -m gpu > $O/pytest_dcn.txt 2>&1
tail -3 $O/pytest_dcn.txt
cd /tmp && export TMPDIR=/tmp
for v in win nowin; do
  if [ $v = nowin ]; then export PT_DCN_WIN=0; fi
  rm -rf /tmp/prof_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/bench.py --stages tsr --no-cpu-baseline --no-extra-legs --steps 4 --warmup 2 > $O/bench_tsr_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_tsr_$v.csv && grep "dcn_" $f | cut -c1-60,150-400
  tail -1 $O/bench_tsr_$v.log | cut -c1-200
done
