#!/bin/bash
# tools/profile_round.sh <tag>: the measurement set committed under profiles/<round>/ (run on the GPU box through gpurun).
#   1. bench.py default (compact line + bench_detail.json)                          -> gpurun_out/<tag>/bench_line.json, bench_detail.json
#   2. bench.py --stages det (BASELINE.json configs[1])                             -> gpurun_out/<tag>/bench_det.json
#   3. rocprofv3 --kernel-trace --stats of the default bench (timed legs only)      -> gpurun_out/<tag>/kernel_stats.csv, gaps.txt
#   4. separate PMC passes: FETCH_SIZE, WRITE_SIZE (-> pmc_summary.json), MFMA busy (-> mfma_busy.json)
#   5. rocprofv3 --marker-trace --kernel-trace of a short run: the roctx stage ranges (pdf_table_amd/trace_ranges.py)   -> marker_summary.txt
set -x
T=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --steps 20 --warmup 8 --detail-out $O/bench_detail.json 2>$O/bench_full.err | tail -1 > $O/bench_line.json
timeout 300 python $R/bench.py --stages det --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --detail-out $O/bench_det_detail.json 2>/dev/null | tail -1 > $O/bench_det.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --detail-out /tmp/d1.json > $O/stats.log 2>&1
KS=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && cp $KS $O/kernel_stats.csv
KT=$(find $O/stats -name "*kernel_trace.csv" | head -1); [ -n "$KT" ] && python $R/tools/trace_gaps.py $KT 30 600 > $O/gaps.txt 2>&1
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-post --detail-out /tmp/d2.json > $O/pmc_$c.log 2>&1
done
mkdir -p $O/pmc_all && cp -r $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_all/ 2>/dev/null
python $R/tools/pmc_summary.py $O/pmc_all $O/pmc_summary.json > /dev/null 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-post --detail-out /tmp/d3.json > $O/pmc_mfma.log 2>&1
python $R/tools/mfma_busy.py $O/pmc_mfma $O/mfma_busy.json > $O/mfma_busy.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_all $O/pmc_mfma
timeout 600 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/marker -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --detail-out /tmp/d4.json > $O/marker.log 2>&1
python $R/tools/marker_summary.py $O/marker > $O/marker_summary.txt 2>&1
rm -rf $O/marker
rm -f $O/*.log
ls -la $O
cut -c1-400 $O/bench_line.json
