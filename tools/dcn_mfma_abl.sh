#!/bin/bash
# tools/dcn_mfma_abl.sh <tag> <variant> ...: per-layer times of dcn_mfma_kernel variants (tools/dcn_variant.sh builds them) over the table-structure stage
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/prof_$v
  lib=$R/tools/scratch/lib_$v.so
  [ $v = base ] && lib=$R/pdf_table_amd/libpdftable_hip.so
  PT_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -- python $R/bench.py --stages tsr --no-cpu-baseline --no-extra-legs --no-post --steps 3 --warmup 1 > $O/abl_$v.log 2>&1
  echo "== $v"; python $R/tools/dcn_by_layer.py /tmp/prof_$v | tee $O/dcn_by_layer_$v.txt
done
