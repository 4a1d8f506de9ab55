"""Per-launch-label time of ONE stage of bench.py's step (HIP events around every launch: pt_profile_enable(1)): what the stage is made of.
usage: python tools/stage_labels.py det [steps] [bf16|bf16x3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    stage = sys.argv[1] if len(sys.argv) > 1 else "det"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    args = bench.parse_args(["--stages", stage, "--no-post"])
    r = bench.HipRunner(args, 0, 0, 1, None)
    if prec == "bf16x3":
        r.eng.set_precision(r.L.PT_PRECISION_BF16X3)
    r.run(4)
    r.sync()
    r.eng.profile_enable(1)
    r.run(steps)
    r.sync()
    labels = r.eng.profile_read_labels()
    r.eng.profile_enable(False)
    tot = sum(v["ms"] for v in labels.values()) / steps
    print(f"{stage} [{prec}]: {tot:.3f} ms of kernels per step ({steps} steps)")
    for lab, v in sorted(labels.items(), key=lambda kv: -kv[1]["ms"]):
        fl = v["flop"] / steps / 1e9
        ms = v["ms"] / steps
        print(f"{ms:8.3f} ms {v['launches'] / steps:6.1f} launches {fl:9.1f} GFLOP {('%.3f of MFMA peak' % (fl / ms / 2.5e3)) if fl > 0 else '':22s} {v['bytes'] / steps / 1e9:7.2f} GB  {lab}")


if __name__ == "__main__":
    main()
