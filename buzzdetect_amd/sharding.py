"""Multi-GPU layout of the analyze path: one process per GPU, recordings dealt round-robin,
one gather of per-window logits to rank 0.

The reference has no distributed mode (threads over one queue, src/analyze.py:218-253); its unit of
independent work is the chunk (src/pipeline/assignments.py:35-41).  Windows never depend on another
chunk's audio (hazard H1: a chunk's last 240 samples are zero padding), so sharding whole files — or
whole chunks of one long file — changes no result.  Nothing but ``[W,13]`` f32 logits (52 B/window)
crosses ranks; weights are replicated.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Item ``i`` belongs to rank ``i mod world_size`` (file-level round-robin)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n_items, world_size))


def owner_of(item: int, world_size: int) -> int:
    return item % world_size


def gather_rows(local: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Gather ``[rows_r, C]`` tensors of differing ``rows_r`` to ``dst``.

    One small all_gather of the row counts, then one padded gather of the payload (RCCL ``gather``
    needs equal shapes).  Returns the per-rank tensors on ``dst`` (trimmed), ``None`` elsewhere.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    rows = int(counts.max().item())
    cols = local.shape[1]
    padded = local
    if local.shape[0] != rows:
        padded = torch.zeros((rows, cols), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    if rank == dst:
        bucket = [torch.empty((rows, cols), dtype=local.dtype, device=local.device) for _ in range(world)]
        dist.gather(padded, bucket, dst=dst, group=group)
        return [b[: int(c)] for b, c in zip(bucket, counts.tolist())]
    dist.gather(padded, None, dst=dst, group=group)
    return None


def interleave_round_robin(per_rank: Sequence[Sequence[torch.Tensor]]) -> List[torch.Tensor]:
    """Undo ``shard_indices``: per_rank[r][j] is item ``r + j * world`` -> items in original order."""
    world = len(per_rank)
    total = sum(len(p) for p in per_rank)
    out: List[torch.Tensor] = []
    for i in range(total):
        out.append(per_rank[i % world][i // world])
    return out
