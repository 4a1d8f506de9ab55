"""tools/marker_summary.py <dir of a rocprofv3 --marker-trace --kernel-trace run>: the roctx stage ranges of the pipeline (pdf_table_amd/trace_ranges.py)
-- calls, host milliseconds inside the range -- and the kernel launches whose dispatch time falls inside each stage range (by correlation of the
range's [start, end] on the host timeline with the kernels' dispatch order is not available in the CSV, so kernels are attributed by START timestamp)."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
mk = glob.glob(os.path.join(d, "**", "*marker_api_trace.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not mk:
    print("no marker trace found under", d)
    sys.exit(0)
ranges = defaultdict(lambda: [0, 0.0])
spans = []
with open(mk[0], newline="") as f:
    for r in csv.DictReader(f):
        name = r.get("Function") or r.get("Message") or r.get("Name") or "?"
        msg = r.get("Message") or ""
        label = msg if msg else name
        try:
            t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        except (KeyError, ValueError):
            continue
        ranges[label][0] += 1
        ranges[label][1] += (t1 - t0) / 1e6
        spans.append((t0, t1, label))
print(f"{'range':44s} {'calls':>6s} {'host ms inside':>15s}")
for k, (n, ms) in sorted(ranges.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44s} {n:6d} {ms:15.2f}")
if kt:
    n_k = sum(1 for _ in open(kt[0])) - 1
    print(f"\n{n_k} kernel dispatches in the same run ({os.path.basename(kt[0])})")
