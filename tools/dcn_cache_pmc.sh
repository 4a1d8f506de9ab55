#!/bin/bash
# tools/dcn_cache_pmc.sh <tag>: where the deformable convolution's gather waits -- vector-cache (TCP), L2 (TCC), address-translation and
# texture-addresser (TA) counters of dcn_fused64_kernel over the table-structure stage alone; separate rocprofv3 --pmc passes
# (--kernel-trace only), per-dispatch means by tools/pmc_kernel.py -> gpurun_out/<tag>/dcn_cache_counters.{json,txt}
T=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/dcn_cpmc/g$i -- python $R/bench.py --stages tsr --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-post > $O/dcn_cpmc_g$i.log 2>&1
  tail -2 $O/dcn_cpmc_g$i.log | cut -c1-200
done
python $R/tools/pmc_kernel.py $O/dcn_cpmc dcn_fused64_kernel $O/dcn_cache_counters.json > $O/dcn_cache_counters.txt 2>&1
rm -rf $O/dcn_cpmc
cat $O/dcn_cache_counters.txt | cut -c1-1500
