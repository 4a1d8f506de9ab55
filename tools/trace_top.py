"""tools/trace_top.py <kernel_trace.csv> <name-substring> [n]: the n longest individual launches of the kernels whose
name contains the substring, with grid size, plus a histogram of time by grid size (measurement aid, GPU box)."""
import csv
import sys
from collections import defaultdict

path, sub = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            rows.append((d, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Kernel_Name"][:60]))
tot = sum(d for d, _, _ in rows)
print(f"{len(rows)} launches, {tot / 1e6:.2f} ms total")
by = defaultdict(lambda: [0, 0])
for d, g, _ in rows:
    by[g][0] += d
    by[g][1] += 1
for g, (d, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:n]:
    print(f"  grid {g:8d}  x{c:4d}  {d / 1e6:8.3f} ms  avg {d / c / 1e3:8.1f} us")
