#!/bin/bash
# tools/profile_x3.sh <tag>: rocprofv3 kernel stats + MFMA-busy of the BF16X3 (tolerance-mode) four-stage step
set -x
T=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x3stats -- python $R/bench.py --precision bf16x3 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/x3_stats.log 2>&1
KS=$(find $O/x3stats -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && cp $KS $O/x3_kernel_stats.csv
rm -rf $O/x3stats
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/x3pmc -- python $R/bench.py --precision bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-post > $O/x3_pmc.log 2>&1
python $R/tools/mfma_busy.py $O/x3pmc $O/x3_mfma_busy.json > /dev/null 2>&1
rm -rf $O/x3pmc
tail -2 $O/x3_stats.log | cut -c1-600
