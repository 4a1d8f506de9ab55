"""tools/mfma_busy.py <dir with *counter_collection.csv> <out.json>: matrix-pipe busy fraction per kernel from one
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
    busy = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / N_XCD * N_SIMD)
(SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

N_XCD, N_SIMD = 8, 1024
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
out = {}
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    out[k] = {"dispatches": len(disp[k]), "gui_active_cycles_per_xcd": gui / N_XCD,
              "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
              "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / N_XCD * N_SIMD)}
tot = sum(v["gui_active_cycles_per_xcd"] for v in out.values())
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_xcd"])[:16]:
    print(f"{v['gui_active_cycles_per_xcd'] / tot * 100:5.1f} % of cycles  mfma busy {v['mfma_busy_frac'] * 100:5.1f} %  x{v['dispatches']:4d}  {k}")
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
