"""What OcrTablePipeline(precision="fp16", layout_precision="fp32") costs over precision="fp16" alone, by launch label: bench.py's f16 engine with the
layout stage's network in the pair mode (LayoutStage.precision = BF16X3).  usage: python tools/layout_pair_cost.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def labels_of(r, eng, steps):
    r.run(3)
    r.sync()
    dt, _ = r.timed(steps, 0)
    eng.profile_enable(1)
    r.run(steps)
    r.sync()
    lab = eng.profile_read_labels()
    eng.profile_enable(False)
    return dt / steps * 1e3, {k: v["ms"] / steps for k, v in lab.items()}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    args = bench.parse_args([])
    r = bench.HipRunner(args, 0, 0, 1, None)
    L = r.L
    from pdf_table_amd.layout_stage import LayoutStage, PicodetConfig
    from pdf_table_amd.pipeline import OcrTablePipeline
    from pdf_table_amd.weights import pack_picodet
    f = r.second_engine("f16")
    with r.on_engine(f):
        ms0, lab0 = labels_of(r, f["eng"], steps)
    f["eng"].load_weights(L.PT_MODEL_PICODET, pack_picodet(r.ysd, 5, x3=True))
    lay = LayoutStage(f["eng"], PicodetConfig(task_type="en"), precision=L.PT_PRECISION_BF16X3)
    f["layout"] = lay
    f["pipe"] = OcrTablePipeline.from_engine(f["eng"], f["stage"], f["rec"], lay, f["tsr"], overlap_rec=False, aux_layout=bool(args.aux_stream),
                                             tsr_on_aux=bool(args.aux_stream), lookahead=int(os.environ.get("PT_PIPE_LOOKAHEAD", "1")))
    with r.on_engine(f):
        ms1, lab1 = labels_of(r, f["eng"], steps)
    if os.environ.get("PT_LAYOUT_AUX"):      # the pair-mode layout net on the auxiliary stream, beside the MFMA-bound stages
        f["pipe"] = OcrTablePipeline.from_engine(f["eng"], f["stage"], f["rec"], lay, f["tsr"], overlap_rec=False, aux_layout=True, tsr_on_aux=False,
                                                 lookahead=int(os.environ.get("PT_PIPE_LOOKAHEAD", "1")))
        with r.on_engine(f):
            r.run(3); r.sync()
            dt, _ = r.timed(steps, 2)
        print(f"layout stage in the pair mode ON THE AUXILIARY STREAM: {dt / steps * 1e3:.2f} ms per step ({64 * steps / dt:.1f} pages/s)")
    print(f"f16 engine: {ms0:.2f} ms per step ({64e3 / ms0:.1f} pages/s), kernels {sum(lab0.values()):.2f} ms;  layout stage in the pair mode: {ms1:.2f} ms "
          f"({64e3 / ms1:.1f} pages/s), kernels {sum(lab1.values()):.2f} ms")
    lay_marks = ("layout", "lcnet", "picodet", "@400x304", "@200x152", "@100x76", "@50x38", "@25x19", "@13x10")
    is_lay = lambda k: any(m in k for m in lay_marks)
    for name, sel in (("layout stage's launches", is_lay), ("all other launches", lambda k: not is_lay(k))):
        print(f"{sum(v for k, v in lab0.items() if sel(k)):8.2f} -> {sum(v for k, v in lab1.items() if sel(k)):8.2f} ms  {name}")
    keys = sorted(set(lab0) | set(lab1), key=lambda k: -abs(lab1.get(k, 0) - lab0.get(k, 0)))
    for k in [k for k in keys if not is_lay(k)][:8]:
        print(f"{lab0.get(k, 0):8.3f} -> {lab1.get(k, 0):8.3f} ms  {k}")
    import re, collections
    grp0, grp1 = collections.defaultdict(float), collections.defaultdict(float)      # the layout stage's launches by kind (sizes dropped)
    for lab, grp in ((lab0, grp0), (lab1, grp1)):
        for k, v in lab.items():
            if is_lay(k):
                grp[re.sub(r"\s*x3$", "", re.sub(r"@\d+x\d+", "@", re.sub(r"\d+->\d+", "N", k)))] += v
    print("layout stage by kind (single pass -> pair mode):")
    for k in sorted(set(grp0) | set(grp1), key=lambda k: -grp1.get(k, 0)):
        print(f"{grp0.get(k, 0):8.3f} -> {grp1.get(k, 0):8.3f} ms  {k}")


if __name__ == "__main__":
    main()
