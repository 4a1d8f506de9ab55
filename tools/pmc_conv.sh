#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <conv_bench args...>   (on the GPU box; writes gpurun_out/pmc_<tag>/)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_BUSY_avr TCC_REQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCP_TA_TCP_STATE_READ_sum"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/$n -o p -- python $R/tools/conv_bench.py "$@" > /dev/null 2>&1 || echo "pmc group $n failed"
done
cd $R
python - "$tag" <<'PY'
import csv, glob, sys, collections
tag=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:34s} avg/dispatch = {sum(vals)/len(vals):16.0f}  (n={len(vals)})")
PY
