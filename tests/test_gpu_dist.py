"""The multi-GPU code path (the reference loops pages serially, cli/main.py:116-144) executed on the ONE GPU a test
box has: a world-1 RCCL process group (backend "nccl"), the weight-blob broadcast onto device memory, pt_weights_load_device, and the
rank != 0 branches of bench.py's HipRunner.  Each case runs in its own process (a process group is process-global state)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(code, env_extra=None, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        print(r.stdout[-3000:])
        print(r.stderr[-6000:])
    assert r.returncode == 0, "child process failed (its output is printed above)"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


_BROADCAST = r"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import torch.distributed as dist
from pdf_table_amd import lib as L
from pdf_table_amd.dist_utils import broadcast_blob
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.rec_stage import build_lines
from pdf_table_amd.synth_pages import make_page
from pdf_table_amd.synth_weights import crnn_state_dict, db_resnet18_state_dict
from pdf_table_amd.weights import pack_crnn, pack_db_resnet18

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
torch.cuda.synchronize()
t_init = time.perf_counter() - t0
blobs = {L.PT_MODEL_DB_RESNET18: pack_db_resnet18(db_resnet18_state_dict(seed=0, text_signal=True), x3=False),
         L.PT_MODEL_CRNN: pack_crnn(crnn_state_dict(seed=1), x3=False)}
a, b = HipEngine(0), HipEngine(0)
t0 = time.perf_counter()
nbytes = 0
for kind, blob in blobs.items():
    t = broadcast_blob(blob, dev)                      # RCCL broadcast: size, then payload, both device tensors
    assert t.is_cuda and t.dtype == torch.uint8 and t.numel() == len(blob)
    assert bytes(t[:4].cpu().numpy()) == b"PTW1" and np.array_equal(t.cpu().numpy(), np.frombuffer(blob, np.uint8))
    a.load_weights_device(kind, t)                     # pt_weights_load_device: device-to-device, no host copy of the payload
    nbytes += len(blob)
    b.load_weights(kind, blob)                         # pt_weights_load: the single-process path
torch.cuda.synchronize()
t_bcast = time.perf_counter() - t0
made = [make_page(i, 1024) for i in range(2)]
pages = torch.from_numpy(np.stack([m[0] for m in made])).to(dev)
pa, ba = a.det_forward(pages, L.PT_DET_PRE_DB_PP, 0.3)
pb, bb = b.det_forward(pages, L.PT_DET_PRE_DB_PP, 0.3)
torch.cuda.synchronize()
assert torch.equal(pa, pb) and torch.equal(ba, bb), "det_forward differs between the broadcast-loaded and the host-loaded engine"
quads = []
for m in made:
    l = m[1]["lines"].astype(np.float64)
    quads.append(np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1))
lines = build_lines(quads)
ra = a.rec_forward(pages, lines, want_maxlogit=True)
rb = b.rec_forward(pages, lines, want_maxlogit=True)
torch.cuda.synchronize()
a.check(); b.check()
ia, ib = ra[0].cpu().numpy(), rb[0].cpu().numpy()
assert ia.shape[0] == sum(len(q) for q in quads) > 50
assert np.array_equal(ia, ib) and np.array_equal(ra[1].cpu().numpy(), rb[1].cpu().numpy()), "rec_forward differs"
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)               # bench.py's max-over-ranks reduction
assert float(t.item()) == 1.5
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"rccl_init_s": t_init, "broadcast_load_s": t_bcast, "blob_bytes": nbytes, "lines": int(ia.shape[0]),
                  "prob_sum": float(pa.double().sum().item())}))
"""


def test_world1_rccl_broadcast_load_device_bit_identical():
    """dist.init_process_group("nccl") + broadcast_blob of two packed blobs onto cuda:0 + load_weights_device: one det_forward and one
    rec_forward bit-identical to an engine loaded through load_weights (bench.py:429-430 is exactly this sequence)"""
    out = _run(_BROADCAST)
    assert out["blob_bytes"] > 40_000_000 and out["lines"] > 50 and out["prob_sum"] > 0
    print(f"RCCL world-1: init {out['rccl_init_s']:.2f} s, broadcast + load of {out['blob_bytes'] / 1e6:.1f} MB {out['broadcast_load_s']:.2f} s")


_RANK1 = r"""
import json, sys
sys.path.insert(0, ".")
import numpy as np, torch
import torch.distributed as dist
import bench
from pdf_table_amd import dist_utils

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
args = bench.parse_args(["--gpus", "2", "--no-extra-legs", "--no-cpu-baseline"])

# rank 0 of a 2-rank job: the real broadcast (world-1 RCCL group), every payload recorded as the tensor a second rank would receive
real = dist_utils.broadcast_blob
sent = []
def recording(blob, device, src=0, group=None):
    assert blob is not None, "rank 0 must pack every blob"
    t = real(blob, device, src, group)
    sent.append(t.clone())
    return t
dist_utils.broadcast_blob = recording
r0 = bench.HipRunner(args, 0, 0, 2, dist)

# rank 1 of the same job on the same device: packs NOTHING (blob is None on every call), receives rank 0's payloads in order
got = []
def replaying(blob, device, src=0, group=None):
    assert blob is None, "a rank other than 0 packed a weight blob"
    t = sent[len(got)]
    got.append(t)
    return t
dist_utils.broadcast_blob = replaying
r1 = bench.HipRunner(args, 1, 0, 2, dist)
assert len(got) == len(sent) >= 5
assert r1.sd is None and r1.csd is None and r1.lsd is None and r1.psd is None and r1.ysd is None      # the first == False branches
assert r0.sd is not None
assert not np.array_equal(r0.pages_np, r1.pages_np)            # rank 1 owns the next shard of the global page batch
c = r1.run(2, count=True)
r1.sync()
r1.eng.check()
c0 = r0.run(2, count=True)
r0.sync()
cfg = r1.config(c, 2)
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"blobs": len(sent), "boxes": cfg["boxes_per_page"], "lines": cfg["text_lines_recognised_per_page"], "tokens": cfg["tokens_per_page"],
                  "cells": cfg["table_cells_per_page"], "tables": cfg["tables_per_page"], "rank0_boxes": c0["boxes"] / (2 * bench.PAGES_PER_STEP),
                  "post_workers": r1.post_workers, "host_cores": r1.host_cores}))
"""


def test_hiprunner_rank1_of_2_branches_execute():
    """bench.py's HipRunner constructed as rank 1 of a 2-rank job (self.sd = None, lambdas that must not pack, load_weights_device of the
    payloads rank 0 broadcast, pages of the second shard, a worker pool of half the cores) runs the four-stage step and finds boxes,
    lines and cells on its own pages"""
    out = _run(_RANK1, {"PT_BENCH_PAGES": "8"})
    assert out["blobs"] >= 5
    assert out["boxes"] > 20 and out["lines"] > 20 and out["tokens"] > 0 and out["cells"] > 0 and out["tables"] >= 1
    assert out["rank0_boxes"] > 20
    assert out["post_workers"] == max(1, min(32, out["host_cores"] // 2))
