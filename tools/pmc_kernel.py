"""tools/pmc_kernel.py <dir with *counter_collection.csv> <kernel substring> [out.json]: per-dispatch means of every counter of one rocprofv3
--pmc pass for the kernels whose name contains the substring, with the derived VALU figures (per wave, per SIMD cycle) the DCN analysis of
DESIGN.md section 9 rests on.  SQ_* counters are summed over the chip; GRBM_GUI_ACTIVE over the 8 XCDs."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
passes = defaultdict(lambda: defaultdict(set))          # a counter that sits in several passes (GRBM_GUI_ACTIVE) is averaged over them
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if sys.argv[2] not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
            passes[k][r["Counter_Name"]].add(f)
out = {}
for k, c in acc.items():
    n = len(disp[k])
    d = {name: v / n / len(passes[k][name]) for name, v in c.items()}
    d["dispatches"] = n
    gui = d.get("GRBM_GUI_ACTIVE", 0.0) / 8
    if gui and "SQ_ACTIVE_INST_VALU" in d:
        # SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (4 clocks); cross-check: SQ_INSTS_VALU x 4 clocks per wave64 instruction
        d["valu_active_frac_of_simd_cycles"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / (gui * 1024)
        if "SQ_INSTS_VALU" in d:
            d["valu_issue_frac_from_inst_count"] = 4.0 * d["SQ_INSTS_VALU"] / (gui * 1024)
    if d.get("SQ_WAVES") and "SQ_INSTS_VALU" in d:
        d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
    if d.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in d:
        d["wait_inst_any_frac_of_wave_cycles"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
    if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024)
    out[k] = d
    print(k, json.dumps({kk: (round(vv, 4) if isinstance(vv, float) and vv < 10 else round(vv)) for kk, vv in d.items()}))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
