#!/bin/bash
# tools/dcn_mfma_pmc.sh <tag> [kernel-name-pattern]: issue / wait / LDS / vector-cache counters of the deformable-convolution kernel over the
# table-structure stage alone; separate rocprofv3 --pmc passes (--kernel-trace only), per-dispatch means by tools/pmc_kernel.py
T=${1:-r04}
K=${2:-dcn_mfma_kernel}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$K/g$i -- python $R/bench.py --stages tsr --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-post > $O/pmc_${K}_g$i.log 2>&1
  tail -1 $O/pmc_${K}_g$i.log | cut -c1-120
done
python $R/tools/pmc_kernel.py $O/pmc_$K $K $O/${K}_counters.json > $O/${K}_counters.txt 2>&1
rm -rf $O/pmc_$K
cat $O/${K}_counters.txt | cut -c1-2500
