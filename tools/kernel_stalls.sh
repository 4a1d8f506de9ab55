#!/bin/bash
# tools/kernel_stalls.sh <out.txt> <kernel regex> <command...>: SQ stall / LDS / MFMA counters of the kernels matching <regex> in <command> (on the GPU box).
# Separate --pmc passes restricted to those kernels (--kernel-include-regex: everything else runs uninstrumented); per-kernel averages appended to <out.txt>.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; rx=$2; shift 2
tag=$(echo "$rx" | tr -c 'A-Za-z0-9\n' '_')
echo "=== $rx: $*" >> $out
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  n=$(echo $grp | cut -d' ' -f1)
  (cd $R && timeout 300 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "$rx" --output-format csv -d $R/gpurun_out/pmc_$tag/$n -o p -- "$@" > /dev/null 2>&1) || echo "pmc group $n failed" >> $out
done
cd $R
python - "$tag" "$rx" >> $out <<'PY'
import csv, glob, sys, collections, re
tag, rx = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(rx, r["Kernel_Name"]):
            acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    c = {n: sum(x) / len(x) for n, x in v.items()}
    print(k, f"({len(next(iter(v.values())))} dispatches)")
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print(f"   of the wave cycles: parked (s_waitcnt / barrier) {c.get('SQ_WAIT_ANY', 0) / wc:.3f}  issue stall {c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} "
              f"(LDS issue {c.get('SQ_WAIT_INST_LDS', 0) / wc:.3f})  issuing {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}")
    if c.get("GRBM_GUI_ACTIVE"):
        print(f"   MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}   LDS bank-conflict share of LDS cycles "
              f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, c.get('SQ_LDS_IDX_ACTIVE', 1)):.3f}   LDS active / GUI active per CU "
              f"{c.get('SQ_LDS_IDX_ACTIVE', 0) / (c['GRBM_GUI_ACTIVE'] / 8 * 256):.3f}")
    for n, x in sorted(c.items()):
        print(f"   {n:30s} {x:16.0f}")
PY
