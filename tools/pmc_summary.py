#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) per kernel into JSON.

    python tools/pmc_summary.py <dir with *counter_collection.csv> [out.json]

Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE tallies
128-B requests at 64 B for wide coalesced streaming reads, so the read side is doubled ("fetch_x2") before it is
compared with a byte count.  WRITE_SIZE is left as reported (uncalibrated).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                cn = row.get("Counter_Name") or ""
                try:
                    v = float(row.get("Counter_Value") or 0)
                except ValueError:
                    continue
                a = acc[name.replace("(anonymous namespace)::", "").split("(")[0]][cn]
                a[0] += v
                a[1] += 1
    out = {}
    for k, cs in acc.items():
        e = {}
        for cn, (tot, n) in cs.items():
            e[cn] = {"sum": tot, "dispatches": n, "avg": tot / max(1, n)}
        f = e.get("FETCH_SIZE", {}).get("avg")
        w = e.get("WRITE_SIZE", {}).get("avg")
        if f is not None:
            e["read_bytes_per_launch_corrected"] = f * 1024 * 2
        if w is not None:
            e["write_bytes_per_launch"] = w * 1024
        if f is not None and w is not None:
            e["hbm_bytes_per_launch"] = f * 1024 * 2 + w * 1024
        out[k] = e
    js = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(js)
    print(js[:3000])


if __name__ == "__main__":
    main()
