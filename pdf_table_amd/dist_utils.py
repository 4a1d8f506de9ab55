"""Page sharding + one-off weight broadcast for the multi-GPU path (one process per GPU).

The path shards by page (pages are independent, the reference itself loops them serially,
cli/main.py:116-144): every rank holds a full replica of the weights and owns a contiguous slice of the page
batch.  The only collective is a broadcast of each packed weight blob from rank 0 at start-up (RCCL over xGMI
with backend "nccl"; "gloo" on CPU for the tests).  There is NO collective in steady state; results are
returned per rank (``gather_results`` is a convenience for callers that want them on rank 0).
"""
from __future__ import annotations

import pickle
from typing import List, Optional, Sequence, Tuple

import torch

__all__ = ["shard_range", "broadcast_blob", "gather_results"]


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Static contiguous split of ``n_items`` pages: first ``n % world`` ranks get one extra page."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_blob(blob: Optional[bytes], device: torch.device, src: int = 0, group=None) -> torch.Tensor:
    """Rank ``src`` passes the packed weight blob (bytes); every rank gets it back as a uint8 tensor on
    ``device`` (ready for ``HipEngine.load_weights_device``).  Two broadcasts: the size, then the payload."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    if rank == src:
        if blob is None:
            raise ValueError("source rank must provide the blob")
        payload = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
        size = torch.tensor([payload.numel()], dtype=torch.int64, device=device)
    else:
        size = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(size, src, group=group)
    if rank != src:
        payload = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(payload, src, group=group)
    return payload


def gather_results(local: Sequence, n_total: int, group=None) -> Optional[List]:
    """Collect per-page results (arbitrary picklable objects) on rank 0 in page order.  Not used in the timed
    path of bench.py -- results stay on the rank that produced them, as the north star specifies."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = [None] * world if rank == 0 else None
    dist.gather_object(list(local), out, dst=0, group=group)
    if rank != 0:
        return None
    merged = [x for part in out for x in part]
    if len(merged) != n_total:
        raise RuntimeError(f"gathered {len(merged)} results for {n_total} pages")
    return merged
