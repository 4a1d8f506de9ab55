#!/bin/bash
# tools/ab_env.sh <stages> <ENVVAR> <value...>: bench.py pages/s for each value of one environment switch, twice
R=${GRAFT_REPO_ROOT:-/root/repo}
st=$1; var=$2; shift 2
for i in 1 2; do
  for v in "$@"; do
    r=$(env $var=$v timeout 250 python $R/bench.py --stages $st --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
    echo "$var=$v $r"
  done
done
