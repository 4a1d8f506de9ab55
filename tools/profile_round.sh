#!/bin/bash
# tools/profile_round.sh <tag>: the measurement set committed under profiles/ (run on the GPU box through gpurun).
#   1. bench.py default (det+rec+tsr, with cpu_baseline)            -> gpurun_out/<tag>/bench_full.json
#   2. bench.py --stages det (BASELINE.json configs[1])             -> gpurun_out/<tag>/bench_det.json
#   3. rocprofv3 --kernel-trace --stats of the default bench        -> gpurun_out/<tag>/stats/
#   4. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) + summary   -> gpurun_out/<tag>/pmc_summary.json
set -x
T=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py 2>$O/bench_full.err | tail -1 > $O/bench_full.json
timeout 300 python $R/bench.py --stages det --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_det.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-post > $O/pmc_$c.log 2>&1
done
mkdir -p $O/pmc_all && cp -r $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_all/ 2>/dev/null
python $R/tools/pmc_summary.py $O/pmc_all $O/pmc_summary.json > /dev/null 2>&1
# keep only the small summaries (the raw traces are large)
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -delete
ls -la $O $O/stats/* | head -40
cat $O/bench_full.json | cut -c1-300
