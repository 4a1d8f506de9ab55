"""roctx ranges per pipeline stage ("keep the metric dict shape; add rocprof / roctx ranges per stage").

The reference brackets its stages with wall-clock timers and reports them in ``metric`` (ocr_system_task.py:148-166, 646-660); this engine
keeps that dict and ALSO names the same spans for the profiler: ``rocprofv3 --marker-trace`` shows which launches belong to layout,
text detection, recognition and table structure.  The library is found through ctypes (libroctx64, or the rocprofiler-sdk roctx library);
when neither is present, or PT_ROCTX=0, every call is a no-op -- tracing never changes what runs."""
from __future__ import annotations

import contextlib
import ctypes
import os

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("PT_ROCTX", "1") == "0":
        return None
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.argtypes = []
            lib.roctxRangePop.restype = ctypes.c_int
            _lib = lib
            break
        except (OSError, AttributeError):
            continue
    return _lib


def available() -> bool:
    return _load() is not None


def push(name: str) -> None:
    lib = _load()
    if lib is not None:
        lib.roctxRangePushA(name.encode())


def pop() -> None:
    lib = _load()
    if lib is not None:
        lib.roctxRangePop()


@contextlib.contextmanager
def stage_range(name: str):
    """with stage_range("text_detection"): ... -- a named span on the calling thread (nesting allowed)"""
    push(name)
    try:
        yield
    finally:
        pop()
