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
CPP_SOURCES = ["db_post.cpp"]
HEADERS = ["common.h", os.path.join("..", "..", "include", "pdftable_hip.h")]

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
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        cmd = [hipcc] + FLAGS + lang + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
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
