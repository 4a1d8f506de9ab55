"""world_size-2 gloo run of the multi-GPU plumbing on CPU: weight-blob broadcast + page sharding + gather.
The data path itself has no collective (pages shard)."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from pdf_table_amd.dist_utils import shard_range


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 64, 513):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                got += list(range(lo, hi))
            assert got == list(range(n))
            sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from pdf_table_amd.dist_utils import broadcast_blob, gather_results, shard_range
    from pdf_table_amd.synth_weights import db_resnet18_state_dict
    from pdf_table_amd.weights import pack_db_resnet18
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = pack_db_resnet18(db_resnet18_state_dict(seed=5), x3=False) if rank == 0 else None
    t = broadcast_blob(blob, torch.device("cpu"))
    digest = hashlib.sha256(t.numpy().tobytes()).hexdigest()
    lo, hi = shard_range(11, rank, world)
    local = [{"page": i, "boxes": np.full((i % 3, 8), i, np.float32)} for i in range(lo, hi)]
    merged = gather_results(local, 11)
    if rank == 0:
        assert [m["page"] for m in merged] == list(range(11))
        assert t[:4].numpy().tobytes() == b"PTW1"
    q.put((rank, digest, t.numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2] > 1_000_000
