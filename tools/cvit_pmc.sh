#!/bin/bash
# instruction mix of the ConvNextViT kernels: tools/cvit_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cvit_pmc; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o c -- python $R/tools/cvit_bench.py --lines 512 --steps 1 > $O/p$i.log 2>&1
  python - $O/p$i <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counters", sys.argv[1]); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "mlp_kernel" not in k and "dwconv_ln_kernel<8>" not in k: continue
    k = k[k.index("cvit_"):][:28]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
  rm -rf $O/p$i
done
