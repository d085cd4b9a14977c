"""Utterance-batch data parallelism: one process per GPU, clips sharded across ranks.

Clips are independent end to end (no cross-clip state in the hot path), so the only collectives are
the batch scatter (rank 0 holds the audio) and the gather of token ids -- RCCL over xGMI on the GPUs
(``backend="nccl"`` is RCCL on ROCm), gloo in the CPU tests.  Weights are replicated: every rank loads
the same model file, no broadcast is needed.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int]:
    """(rank, world) from the torchrun environment; world == 1 needs no process group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if device is not None and device.type == "cuda":
            kw["device_id"] = device
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), **kw)
    return rank, world


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split; the first (n_items % world) ranks take one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scatter_clips(clips: list[np.ndarray] | None, world: int, rank: int, device: torch.device) -> tuple[torch.Tensor, list[int]]:
    """Rank 0 passes the full clip list, the others None.  Every rank returns (padded [n_local, max_len]
    float32 tensor on ``device``, true lengths).  One scatter of lengths + one scatter of samples."""
    if world == 1:
        assert clips is not None
        lens = [int(c.shape[0]) for c in clips]
        buf = np.zeros((len(clips), max(lens)), np.float32)
        for i, c in enumerate(clips):
            buf[i, : lens[i]] = c
        return torch.from_numpy(buf).to(device), lens
    meta = [None]
    if rank == 0:
        lens_all = [int(c.shape[0]) for c in clips]
        meta = [(len(clips), max(lens_all))]
    dist.broadcast_object_list(meta, src=0)
    n, max_len = meta[0]
    counts = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]
    per = max(counts)  # equal-sized scatter chunks (padded with empty clips)
    lens_local = torch.zeros(per, dtype=torch.int64, device=device)
    audio_local = torch.zeros((per, max_len), dtype=torch.float32, device=device)
    if rank == 0:
        lens_chunks, audio_chunks = [], []
        for r in range(world):
            lo, hi = shard_bounds(n, r, world)
            lt = torch.zeros(per, dtype=torch.int64)
            at = torch.zeros((per, max_len), dtype=torch.float32)
            for j, i in enumerate(range(lo, hi)):
                lt[j] = lens_all[i]
                at[j, : lens_all[i]] = torch.from_numpy(np.ascontiguousarray(clips[i], dtype=np.float32))
            lens_chunks.append(lt.to(device))
            audio_chunks.append(at.to(device))
        dist.scatter(lens_local, lens_chunks, src=0)
        dist.scatter(audio_local, audio_chunks, src=0)
    else:
        dist.scatter(lens_local, None, src=0)
        dist.scatter(audio_local, None, src=0)
    k = counts[rank]
    return audio_local[:k], [int(v) for v in lens_local[:k].tolist()]


def gather_tokens(local: list[list[int]], n_total: int, world: int, rank: int, device: torch.device) -> list[list[int]]:
    """All ranks receive the token lists of every clip, in global clip order."""
    if world == 1:
        return local
    counts = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    per = max(counts)
    width = torch.tensor([max((len(t) for t in local), default=0)], dtype=torch.int64, device=device)
    dist.all_reduce(width, op=dist.ReduceOp.MAX)
    w = int(width.item())
    host = np.full((per, w + 1), -1, dtype=np.int32)  # column 0 = length; packed on the host, ONE copy to the device
    for i, t in enumerate(local):
        host[i, 0] = len(t)
        if t:
            host[i, 1 : 1 + len(t)] = t
    mine = torch.from_numpy(host).to(device)
    out = torch.empty((world * per, w + 1), dtype=torch.int32, device=device)  # concatenated along dim 0
    dist.all_gather_into_tensor(out, mine)
    out = out.view(world, per, w + 1).cpu().numpy()
    res: list[list[int]] = []
    for r in range(world):
        for i in range(counts[r]):
            n = int(out[r, i, 0])
            res.append(out[r, i, 1 : 1 + n].tolist())
    return res


def max_over_ranks(value: float, world: int, device: torch.device) -> float:
    if world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
