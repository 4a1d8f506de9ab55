#!/usr/bin/env python
"""Which kernels make an MFMA wait for an LDS read issued just in front of it?  (No GPU needed.)
Compiles pdf_table_amd/csrc/*.hip to gfx950 assembly (bf16 instantiation) and reports, per kernel with >= 16 MFMAs, the share of v_mfma instructions that
have `s_waitcnt ... lgkmcnt(0)` within the two instructions in front of them -- the signature of ds_reads sunk to their uses (hipcc does that whenever the
source does not pin the order with sched_group_barrier, or a FLAT-encoded access in flight makes its wait-count pass give up on counting) -- and the number
of FLAT-encoded memory instructions.  Round 6 found the CTC classifier (0.39 -> 0.02 with cls_argmax_dma_kernel), the cluster LSTM (1.00 -> 0.03; 96 FLAT
loads) and the DB head (64 FLAT loads from LDS) this way.  tests/test_isa_schedule.py pins the hot kernels' readings.
    python tools/isa_wait_audit.py [substring of the kernel names to keep]"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pdf_table_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def audit(files=None):
    """-> {demangled kernel name: (n_mfma, n_waiting, n_flat)} over the given .hip files (default: all of csrc/)"""
    files = files or sorted(glob.glob(os.path.join(SRC, "*.hip")))
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for f in files:
            asm = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
            procs.append((asm, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + SRC, "-x", "hip", "-S",
                                                 "--cuda-device-only", "-o", asm, f], stderr=subprocess.DEVNULL)))
        rows = []
        for asm, p in procs:
            if p.wait() != 0 or not os.path.exists(asm):
                raise RuntimeError("hipcc failed on " + asm)
            name, mf, waits, flat, prev = None, 0, 0, 0, []
            for ln in open(asm):
                m = re.match(r"^(_Z[\w]+):", ln)
                if m:
                    if name:
                        rows.append((name, mf, waits, flat))
                    name, mf, waits, flat, prev = m.group(1), 0, 0, 0, []
                    continue
                t = ln.strip()
                if not t or t[0] in ";.":
                    continue
                if t.startswith("v_mfma"):
                    mf += 1
                    waits += any(q.startswith("s_waitcnt") and "lgkmcnt(0)" in q for q in prev[-2:])
                if t.startswith(("flat_load", "flat_store", "flat_atomic")):
                    flat += 1
                prev = (prev + [t])[-3:]
            if name:
                rows.append((name, mf, waits, flat))
    names = [r[0] for r in rows]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
    for (name, mf, w, fl), d in zip(rows, dem):
        d = d.replace("(anonymous namespace)::", "")
        out[re.sub(r"\(.*$", "", d) if "(" in d else d] = (mf, w, fl)
    return out


if __name__ == "__main__":
    keep = sys.argv[1] if len(sys.argv) > 1 else ""
    res = audit()
    for name, (mf, w, fl) in sorted(res.items(), key=lambda kv: -(kv[1][1] / max(1, kv[1][0]))):
        if (mf >= 16 or fl) and keep in name:
            print(f"{w / max(1, mf):5.2f}  {mf:4d} MFMAs  {fl:3d} FLAT  {name[-110:]}")
