"""Clip-axis sharding over the GPUs of one node (one process per GPU, RCCL through
``torch.distributed`` backend "nccl"; "gloo" on CPU for tests).

Every Q-Former item (= one temporal position of one video, reference
``models/xinstructblip.py:262-275``) is independent, so a long video's items are split into
contiguous blocks, one per rank, with replicated weights and no collective inside the Q-Former.
The only exchange is one all-gather of the per-clip query embeddings ``[N/world, 32, 768]`` (and
the [CLS] text vectors) before scoring, so that every rank scores the time-ordered whole.
The reference itself has no such collective (inference is single-GPU, DDP only in training).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world(group=None) -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of ``n`` items for ``rank``; the first ``n % world`` ranks get one
    extra item, so blocks stay in temporal order and differ by at most one item."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world_size: int) -> List[int]:
    return [shard_range(n, r, world_size)[1] - shard_range(n, r, world_size)[0] for r in range(world_size)]


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Concatenate every rank's ``[n_local, ...]`` block along dim 0 in rank order -> ``[n_total, ...]``.

    Equal shards use ONE ``all_gather_into_tensor`` (a single RCCL all-gather; on the 8 fully
    connected xGMI peers of an MI355X node this is a direct exchange of 1.5-3 MB per rank, latency
    bound).  Ragged shards are padded to the largest block and trimmed after the gather."""
    rank, ws = world(group)
    if ws == 1:
        return local
    sizes = shard_sizes(n_total, ws)
    mx = max(sizes)
    tail = tuple(local.shape[1:])
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: local block has {local.shape[0]} rows, expected {sizes[rank]}")
    if min(sizes) == mx:
        out = torch.empty((n_total,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((ws * mx,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = [buf[r * mx: r * mx + sizes[r]] for r in range(ws)]
    return torch.cat(parts, dim=0)


def all_gather_packed(locals_: List[torch.Tensor], n_total: int, group=None) -> List[torch.Tensor]:
    """``all_gather_rows`` of several per-item tensors (same leading dimension) with ONE collective: the rows are
    packed side by side into ``[n_local, sum(widths)]``, gathered, and split back into ``[n_total, ...]`` views.
    Four latency-bound RCCL calls per step (query embeddings and [CLS] vectors of two modalities) become one."""
    rank, ws = world(group)
    if ws == 1:
        return list(locals_)
    n_local = locals_[0].shape[0]
    dtype = locals_[0].dtype
    flat = [t.reshape(n_local, -1).to(dtype) for t in locals_]
    widths = [f.shape[1] for f in flat]
    packed = torch.cat(flat, dim=1) if len(flat) > 1 else flat[0].contiguous()
    full = all_gather_rows(packed, n_total, group)
    out, off = [], 0
    for t, wdt in zip(locals_, widths):
        out.append(full[:, off: off + wdt].reshape((n_total,) + tuple(t.shape[1:])).to(t.dtype))
        off += wdt
    return out
