"""Process-wide side streams, one per (device, role).

HIP maps streams onto a handful of hardware queues round-robin in creation order, and torch hands out its pooled streams the same way.  Every
stage used to create its own copy / side stream: the SECOND pipeline of a process got the next streams of the pool, i.e. another stream -> queue
assignment, and ran every kernel ~50 us later than the first pipeline's (round 6, MI355X: four engines in one process, same step: 90.2 / 100.8 /
90.1 / 100.6 ms -- even-numbered engines slow wherever their memory lay; bench.py's f16 leg, a second engine, read 0.91 x the bf16 headline).
A copy stream that shares a hardware queue with the compute stream puts its event waits in front of that queue's kernels.  With one stream per
role for the whole process every pipeline runs in the assignment the first one got."""
from __future__ import annotations

import threading

import torch

_lock = threading.Lock()
_streams = {}


def shared_stream(device, role: str, priority: int = 0) -> "torch.cuda.Stream":
    """the process's stream for `role` ("det_side", "layout_copy", "tsr_copy", "rec", "aux", "h2d", "table") on `device`, created at first use"""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    with _lock:
        s = _streams.get(key)
        if s is None:
            s = _streams[key] = torch.cuda.Stream(device=dev, priority=priority)
        return s
