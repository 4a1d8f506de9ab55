#!/bin/bash
# tools/profile_r02.sh <tag> [stages...]: per-kernel stats (rocprofv3 --kernel-trace --stats) + launch-gap attribution of the
# bench for each stage set (default: "det" and the four-stage default), on the GPU box through gpurun.
T=${1:-r02}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=("$@")
[ ${#SETS[@]} -eq 0 ] && SETS=("det" "layout,det,rec,tsr")
for S in "${SETS[@]}"; do
  N=$(echo $S | tr ',' '_')
  PT_BENCH_TRACE=1 timeout 600 python $R/bench.py --stages $S --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 2> $O/bench_$N.err | tail -1 > $O/bench_$N.json
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$N -- python $R/bench.py --stages $S --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/stats_$N.log 2>&1
  KT=$(find $O/stats_$N -name "*kernel_trace.csv" | head -1)
  KS=$(find $O/stats_$N -name "*kernel_stats.csv" | head -1)
  [ -n "$KS" ] && cp $KS $O/kernel_stats_$N.csv
  [ -n "$KT" ] && python $R/tools/trace_gaps.py $KT 30 600 > $O/gaps_$N.txt 2>&1
  rm -rf $O/stats_$N
done
ls -la $O
