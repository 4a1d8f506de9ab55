"""Build libpdftable_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m pdf_table_amd.build [--force]

The library is written next to this package (``pdf_table_amd/libpdftable_hip.so``) so that it
travels with the source tree to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpdftable_hip.so")
STAMP = LIB + ".stamp"

HIP_SOURCES = ["conv_igemm.hip", "det_kernels.hip", "db_model.hip", "rec_kernels.hip", "crnn_model.hip",
               "lore_kernels.hip", "lore_model.hip", "lore_decode.hip", "lore_processor.hip", "layout_kernels.hip", "layout_model.hip", "dbnas_model.hip", "cls_kernels.hip",
               "graph_ops.hip", "cvit_model.hip", "mtl_model.hip", "mtl_decoder.hip", "c_api.hip"]
# compiled ONCE: host-only post-processing and the format-independent corner of the ABI
CPP_SOURCES = ["db_post.cpp", "api_common.cpp"]
HEADERS = ["common.h", "act16.h", os.path.join("..", "..", "include", "pdftable_hip.h")]
# Every HIP translation unit is compiled once per 16-bit activation format (csrc/act16.h): namespace pt_bf16, and namespace pt_f16 with -DPT_ACT_F16=1
FORMATS = [("bf16", []), ("f16", ["-DPT_ACT_F16=1"])]

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _sources():
    return [s for s in HIP_SOURCES + CPP_SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _digest() -> str:
    h = hashlib.sha256()
    for f in _sources() + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(FORMATS).encode())
    # this file generates api_dispatch.cpp (dispatch_source) and decides flags and sources: a change here must rebuild the library too
    with open(os.path.abspath(__file__), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def lint_kernel_enter() -> None:
    """Every __global__ kernel must open with a16_kernel_enter() (csrc/act16.h: MODE.FP16_OVFL for the half format's saturating stores).  Nothing in the
    language enforces it, so the build does: per translation unit, as many a16_kernel_enter() calls as __global__ definitions."""
    import re
    bad = []
    for src in HIP_SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            txt = re.sub(r"//[^\n]*", "", f.read())
        n_k = len(re.findall(r"\b__global__\b", txt))
        n_e = len(re.findall(r"\ba16_kernel_enter\s*\(\s*\)\s*;", txt))
        if n_k != n_e:
            bad.append(f"{src}: {n_k} __global__ kernels, {n_e} a16_kernel_enter() calls")
    if bad:
        raise RuntimeError("kernels without a16_kernel_enter() (csrc/act16.h): " + "; ".join(bad))


def lint_dispatch() -> None:
    """Every pt_* entry point the Python binding declares (lib.py) must have an exported definition: a generated dispatcher (the regex parser in
    _prototypes() silently skips a header declaration it cannot match) or a compile-once definition in CPP_SOURCES."""
    import re
    with open(os.path.join(HERE, "lib.py")) as f:
        wanted = set(re.findall(r"\b(pt_[a-z0-9_]+)\b", f.read()))
    have = {name for _, name, _ in _prototypes()}
    for src in CPP_SOURCES:
        with open(os.path.join(CSRC, src)) as f:
            have |= set(re.findall(r"^(?:int|void|const char\*)\s+(pt_[a-z0-9_]+)\s*\(", f.read(), flags=re.M))
    with open(os.path.join(CSRC, HEADERS[-1])) as f:
        declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", f.read()))
    missing = sorted(n for n in wanted & declared if n not in have)
    if missing:
        raise RuntimeError("entry points declared in include/pdftable_hip.h and bound in lib.py but without a dispatcher: " + ", ".join(missing))


def _prototypes():
    """(return type, name, parameter text) of every function include/pdftable_hip.h declares."""
    import re
    with open(os.path.join(CSRC, HEADERS[-1])) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    return re.findall(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_]\w*(?:\s*\*)?)\s+(pt_[a-z0-9_]+)\s*\(([^;{}()]*)\)\s*;", txt, flags=re.S)


def dispatch_source() -> str:
    """api_dispatch.cpp: the exported C symbols.  Every entry point that c_api.hip / graph_ops.hip define (once per format namespace) gets a
    dispatcher with the header's exact signature; the namespace is chosen by pt_engine::precision when the first parameter is the engine,
    pt_bf16 otherwise (plans and sizes do not depend on the format).  Functions defined in CPP_SOURCES are exported by those files."""
    import re
    once = set()
    for src in CPP_SOURCES:
        with open(os.path.join(CSRC, src)) as f:
            once |= set(re.findall(r"^(?:int|void|const char\*)\s+(pt_[a-z0-9_]+)\s*\(", f.read(), flags=re.M))
    decl, body = [], []
    for ret, name, params in _prototypes():
        if name in once:
            continue
        params = re.sub(r"\s+", " ", params.strip())
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        names = [re.match(r".*?([A-Za-z_]\w*)\s*(?:\[[^\]]*\])?$", p).group(1) for p in plist]
        decl.append(f"{ret} {name}({params});")
        call = f"{name}({', '.join(names)})"
        if plist and re.match(r"(const\s+)?pt_engine\s*\*\s*\w+$", plist[0]):
            e = names[0]
            expr = f"({e} && {e}->precision == PT_PRECISION_F16) ? pt_f16::api::{call} : pt_bf16::api::{call}"
        else:
            expr = f"pt_bf16::api::{call}"
        body.append(f"{ret} {name}({params}) {{ {'' if ret == 'void' else 'return '}{expr}; }}")
    return ("// api_dispatch.cpp -- GENERATED by pdf_table_amd/build.py from include/pdftable_hip.h; not a source file.\n"
            "#include \"common.h\"\n\n"
            "namespace pt_bf16 { namespace api {\n" + "\n".join(decl) + "\n} }\n"
            "namespace pt_f16 { namespace api {\n" + "\n".join(decl) + "\n} }\n\n"
            "extern \"C\" {\n" + "\n".join(body) + "\n}\n")


def build(force: bool = False, verbose: bool = True) -> str:
    lint_kernel_enter()
    lint_dispatch()
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    jobs = []
    for src in _sources():
        hip = src.endswith(".hip")
        for fmt, defs in (FORMATS if hip else [("", [])]):
            obj = os.path.join(objdir, os.path.splitext(src)[0] + ("_" + fmt if fmt else "") + ".o")
            jobs.append((src + (" [" + fmt + "]" if fmt else ""), [hipcc] + FLAGS + defs + (["-x", "hip"] if hip else []) + ["-c", os.path.join(CSRC, src), "-o", obj]))
            objs.append(obj)
    disp = os.path.join(objdir, "api_dispatch.cpp")
    with open(disp, "w") as f:
        f.write(dispatch_source())
    jobs.append(("api_dispatch.cpp", [hipcc] + FLAGS + ["-I", CSRC, "-x", "hip", "-c", disp, "-o", os.path.join(objdir, "api_dispatch.o")]))
    objs.append(os.path.join(objdir, "api_dispatch.o"))
    # biggest translation units first, at most one compiler per core (each format doubles the job count)
    jobs.sort(key=lambda j: -os.path.getsize(j[1][-3]))
    procs = []
    pending = list(jobs)
    running = []
    width = max(1, min(len(jobs), int(os.environ.get("PT_BUILD_JOBS", str(os.cpu_count() or 8)))))
    while pending or running:
        while pending and len(running) < width:
            src, cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        src, p = running.pop(0)
        out, _ = p.communicate()
        procs.append((src, p, out))
    failed = False
    for src, p, out in procs:
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"[build] {src} FAILED", file=sys.stderr)
            if not verbose:
                print(out, file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
