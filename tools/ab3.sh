#!/bin/bash
# tools/ab3.sh: steady-state (600 launches) conv micro-benchmark of the dispatch variants of the in-tree library
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "8 240 240 64 64" "8 240 240 256 64" "8 120 120 128 128" "8 60 60 256 256" "8 30 30 512 512" "64 8 160 256 256" "64 4 160 512 512"; do
  for v in 0 2 3; do
    PT_CONV_VARIANT=$v python $R/tools/conv_bench.py $shape 3 1 ${ITERS:-600} 2>/dev/null | sed "s/^/v$v: /"
  done
done
