"""tools/trace_context.py <kernel_trace.csv> [n] [k]: the n largest GPU idle gaps of a rocprofv3 kernel trace with the k
kernels before and after each (name, queue, duration) -- to see which stage the queue ran dry in front of."""
import csv
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-56:], r.get("Queue_Id", "?")))
rows.sort()
gaps = []
end = rows[0][1]
for i in range(1, len(rows)):
    if rows[i][0] > end:
        gaps.append((rows[i][0] - end, i))
    end = max(end, rows[i][1])
t0 = rows[0][0]
for g, i in sorted(gaps, reverse=True)[:n]:
    print(f"--- idle {g / 1e6:.2f} ms before kernel #{i} at t = {(rows[i][0] - t0) / 1e6:.1f} ms")
    for j in range(max(0, i - k), min(len(rows), i + k)):
        s, e, name, q = rows[j]
        print(f"   {'>>' if j == i else '  '} t={(s - t0) / 1e6:9.2f} ms  {(e - s) / 1e3:9.1f} us  q{q}  {name}")
