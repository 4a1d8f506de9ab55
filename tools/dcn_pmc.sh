#!/bin/bash
# tools/dcn_pmc.sh <tag>: counter evidence for the deformable-conv kernels: separate rocprofv3 --pmc passes over the
# table-structure stage alone, summarised per dispatch by tools/pmc_kernel.py -> gpurun_out/<tag>/dcn_counters.json
set -x
T=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/dcn_pmc/g$i -- python $R/bench.py --stages tsr --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-post > $O/dcn_pmc_g$i.log 2>&1
done
python $R/tools/pmc_kernel.py $O/dcn_pmc dcn_fused64_kernel $O/dcn_counters.json > $O/dcn_counters.txt 2>&1
rm -rf $O/dcn_pmc
cat $O/dcn_counters.txt | cut -c1-900
