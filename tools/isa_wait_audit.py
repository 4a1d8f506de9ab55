#!/usr/bin/env python
"""Which kernels make an MFMA wait for an LDS read issued just in front of it?  (No GPU needed.)
Compiles every pdf_table_amd/csrc/*.hip to gfx950 assembly (bf16 instantiation) and prints, per kernel with >= 16 MFMAs, the share of v_mfma instructions that
have `s_waitcnt ... lgkmcnt(0)` within the two instructions in front of them -- the signature of ds_reads sunk to their uses (hipcc does that whenever the
source does not pin the order with sched_group_barrier, or a FLAT-encoded access in flight makes its wait-count pass give up on counting).  Round 6 found the
CTC classifier (0.39 -> 0.02 after cls_argmax_dma_kernel) and the cluster LSTM (1.00 -> 0.02) this way.
    python tools/isa_wait_audit.py [substring of the kernel names to keep]"""
import glob, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "pdf_table_amd", "csrc")
keep = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
with tempfile.TemporaryDirectory() as tmp:
    procs = []
    for f in sorted(glob.glob(os.path.join(src, "*.hip"))):
        out = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
        procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + src, "-x", "hip",
                                              "-S", "--cuda-device-only", "-o", out, f], stderr=subprocess.DEVNULL)))
    for out, p in procs:
        if p.wait() != 0 or not os.path.exists(out):
            continue
        name, mf, waits, prev = None, 0, 0, []
        for ln in open(out):
            m = re.match(r"^(_Z[\w]+):", ln)
            if m:
                if name and mf:
                    rows.append((name, mf, waits))
                name, mf, waits, prev = m.group(1), 0, 0, []
                continue
            t = ln.strip()
            if not t or t[0] in ";.":
                continue
            if t.startswith("v_mfma"):
                mf += 1
                waits += any(q.startswith("s_waitcnt") and "lgkmcnt(0)" in q for q in prev[-2:])
            prev = (prev + [t])[-3:]
        if name and mf:
            rows.append((name, mf, waits))
seen = set()
for name, mf, w in sorted(rows, key=lambda r: -r[2] / r[1]):
    if mf >= 16 and keep in name and name not in seen:
        seen.add(name)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{w / mf:5.2f}  {mf:4d} MFMAs  {dem[-100:]}")
