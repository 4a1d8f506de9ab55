#!/bin/bash
# tools/lstm_abl.sh: lstm_dir_kernel average duration under the ablation builds tools/scratch/lib_lstm{1..4}.so
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v != base ]; then export PT_LIB_PATH=$R/tools/scratch/lib_$v.so; fi
  rm -rf /tmp/prof_$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/bench.py --stages rec --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then grep "lstm_" "$f" | sed "s/.*)\",//" | cut -d, -f1-3 | sed "s/^/$v: /"; else echo "$v: no stats"; fi
done
