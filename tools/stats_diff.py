"""tools/stats_diff.py <a_kernel_stats.csv> <b_kernel_stats.csv> [steps_a] [steps_b] [n]: per-kernel total time per step of two rocprofv3 --stats runs side by side"""
import csv
import sys


def load(path, steps):
    d = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            d[r["Name"].split("(")[0][-60:]] = (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps)
    return d


a, b = load(sys.argv[1], float(sys.argv[3]) if len(sys.argv) > 3 else 1.0), load(sys.argv[2], float(sys.argv[4]) if len(sys.argv) > 4 else 1.0)
n = int(sys.argv[5]) if len(sys.argv) > 5 else 30
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[0] - b.get(k, (0, 0))[0]))
print(f"{'kernel':60s} {'a ms/step':>10s} {'calls':>7s} {'b ms/step':>10s} {'calls':>7s} {'a - b':>8s}")
for k in keys[:n]:
    x, y = a.get(k, (0, 0)), b.get(k, (0, 0))
    print(f"{k:60s} {x[0]:10.2f} {x[1]:7.1f} {y[0]:10.2f} {y[1]:7.1f} {x[0] - y[0]:8.2f}")
print(f"{'TOTAL':60s} {sum(v[0] for v in a.values()):10.2f} {'':7s} {sum(v[0] for v in b.values()):10.2f}")
