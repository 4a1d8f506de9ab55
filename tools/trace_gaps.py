"""tools/trace_gaps.py <kernel_trace.csv> [n] [window_ms]: where the GPU idles -- gaps between consecutive kernels (start
minus the latest end so far), aggregated by (previous kernel -> next kernel), over the last window_ms of the trace (the
steady-state steps; default 800).  Under rocprofv3 every launch gap is inflated to ~10 us: read the large ones."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
win = float(sys.argv[3]) if len(sys.argv) > 3 else 800.0
rows = [r for r in rows if r[0] >= t1 - int(win * 1e6)]
busy = 0
end = rows[0][0]
gaps = defaultdict(lambda: [0, 0, 0])
hist = defaultdict(lambda: [0, 0])
for i, (s, e, name) in enumerate(rows):
    if s > end:
        g = s - end
        key = (rows[i - 1][2] if i else "-", name)
        gaps[key][0] += g
        gaps[key][1] += 1
        gaps[key][2] = max(gaps[key][2], g)
        b = "<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"
        hist[b][0] += g
        hist[b][1] += 1
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    end = max(end, e)
span = rows[-1][1] - rows[0][0]
print(f"span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms, {len(rows)} kernels")
for b, (g, c) in sorted(hist.items(), key=lambda kv: -kv[1][0]):
    print(f"  gaps {b:7s} x{c:6d}  {g / 1e6:8.2f} ms")
for (a, b), (g, c, m) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:n]:
    print(f"  {g / 1e6:7.2f} ms x{c:4d} max {m / 1e3:8.1f} us  {a} -> {b}")
