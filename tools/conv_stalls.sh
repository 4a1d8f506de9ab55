#!/bin/bash
# tools/conv_stalls.sh <out.txt> <PT_CONV_PIPE value> <conv_harness args...>: SQ stall counters of one conv launch configuration (on the GPU box).
# Separate --pmc passes (kernel-trace only) over the Python-free harness tools/conv_harness; per-kernel averages are appended to <out.txt>.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; pipe=$2; shift 2
tag=$(echo "p${pipe}_$*" | tr ' ' '_')
export PT_CONV_PIPE=$pipe
echo "=== PT_CONV_PIPE=$pipe conv_harness $*" >> $out
$R/tools/conv_harness "$@" >> $out 2>&1
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/$n -o p -- $R/tools/conv_harness "$@" > /dev/null 2>&1 || echo "pmc group $n failed" >> $out
done
cd $R
python - "$tag" >> $out <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    c = {n: sum(x) / len(x) for n, x in v.items()}
    print(k)
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print(f"   of the wave cycles: parked (s_waitcnt / barrier) {c.get('SQ_WAIT_ANY', 0) / wc:.3f}  issue stall {c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} "
              f"(LDS issue {c.get('SQ_WAIT_INST_LDS', 0) / wc:.3f})  issuing {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}")
    if c.get("GRBM_GUI_ACTIVE"):
        print(f"   MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}   LDS bank-conflict share of LDS cycles "
              f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, c.get('SQ_LDS_IDX_ACTIVE', 1)):.3f}")
    for n, x in sorted(c.items()):
        print(f"   {n:30s} {x:16.0f}")
PY
