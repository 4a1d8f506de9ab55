#!/bin/bash
# tools/lds_conflicts.sh <out.txt> <command...>: LDS activity / bank-conflict share and MFMA-busy share of EVERY kernel of <command> (one --pmc pass, on the GPU box)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift
rm -rf $R/gpurun_out/pmc_lds
(cd $R && timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lds -o p -- "$@" > /dev/null 2>&1) || echo "pmc pass failed" >> $out
cd $R
python - >> $out <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))
print(f"{'GUI cycles':>14s} {'calls':>6s} {'MFMA busy':>9s} {'LDS active':>10s} {'conflict share':>14s}  kernel")
for k, c in rows[:45]:
    g = c.get("GRBM_GUI_ACTIVE", 0)
    if not g:
        continue
    print(f"{g:14.0f} {n[k]:6d} {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (g / 8 * 1024):9.3f} {c.get('SQ_LDS_IDX_ACTIVE', 0) / (g / 8 * 256):10.3f} "
          f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 0)):14.3f}  {k}")
PY
