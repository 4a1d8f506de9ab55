#!/bin/bash
# tools/ab2.sh <variant names...>: conv micro-benchmark of the v1 kernel across experimental library builds
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "8 240 240 256 64" "8 60 60 256 256" "8 30 30 512 512"; do
  PT_CONV_VARIANT=0 python $R/tools/conv_bench.py $shape 3 1 600 2>/dev/null | sed 's/^/base: /'
  for v in "$@"; do
    PT_CONV_VARIANT=0 PT_LIB_PATH=$R/tools/scratch/lib_$v.so python $R/tools/conv_bench.py $shape 3 1 600 2>/dev/null | sed "s/^/$v: /"
  done
done
