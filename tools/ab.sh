#!/bin/bash
# tools/ab.sh <variant name>: A/B the conv micro-benchmark between the in-tree library and tools/scratch/lib_<name>.so
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "8 240 240 64 64" "8 240 240 256 64" "8 120 120 128 128" "8 60 60 256 256" "8 30 30 512 512"; do
  for rep in 1 2; do
    PT_CONV_VARIANT=0 python $R/tools/conv_bench.py $shape 3 1 30 2>/dev/null | sed 's/^/base: /'
    PT_CONV_VARIANT=0 PT_LIB_PATH=$R/tools/scratch/lib_$1.so python $R/tools/conv_bench.py $shape 3 1 30 2>/dev/null | sed "s/^/$1: /"
  done
done
