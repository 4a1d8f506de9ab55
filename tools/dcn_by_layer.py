"""tools/dcn_by_layer.py <rocprofv3 output dir>: mean duration of the deformable-convolution launches per (kernel, grid) -- i.e. per layer shape --
from a --kernel-trace run of the table-structure stage (the kernel_stats summary mixes the 16 DCN layers of a DLA-34 up-sampler)."""
import csv
import glob
import sys
from collections import defaultdict

rows = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "dcn_" not in n:
            continue
        short = n[n.index("dcn_"):].split("(")[0]
        key = (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
        rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k in sorted(rows, key=lambda k: -sum(rows[k])):
    v = rows[k]
    tot += sum(v)
    print(f"{k[0]:32s} tiles {k[1]:7d} x {k[2]}  launches {len(v):4d}  mean {sum(v) / len(v):9.1f} us  total {sum(v) / 1e3:8.2f} ms")
print(f"all deformable convolutions: {tot / 1e3:.2f} ms")
