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


def force_group() -> bool:
    """MSH_DIST_FORCE_GROUP=1: create the process group and take the collective path even at world == 1 -- the way to run
    the RCCL init / scatter / all-gather / all-reduce code on a box with ONE GPU (tests/test_gpu_dist.py,
    `MSH_DIST_FORCE_GROUP=1 python bench.py`).  Without it a one-rank run touches no collective at all."""
    return os.environ.get("MSH_DIST_FORCE_GROUP", "") == "1"


def use_collectives(world: int) -> bool:
    return world > 1 or (force_group() and dist.is_initialized())


def init_from_env(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int]:
    """(rank, world) from the torchrun environment; world == 1 needs no process group (unless MSH_DIST_FORCE_GROUP=1)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force_group()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        kw = {}
        if device is not None and device.type == "cuda":
            kw["device_id"] = device
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), **kw)
    return rank, world


def rank_cpus(local_rank: int, local_world: int, allowed: list[int] | None = None, gpu_numa_node: int | None = None,
              node_cpus: dict[int, list[int]] | None = None) -> list[int]:
    """The CPUs one rank of an N-rank node should keep its host threads on (the engine's lane threads, the device VAD's gather
    threads and the pinned staging blocks they first-touch all inherit the process's affinity mask).  Preference: the CPUs of
    the NUMA node the rank's GPU hangs off (sysfs), shared evenly between the ranks whose GPUs sit on that node; without
    topology information an even contiguous split of the allowed CPUs (on the usual two-socket boxes GPUs 0..N/2-1 and the
    lower half of the CPU numbering share a socket).  Pure function of its arguments: tested on the CPU."""
    cpus = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    if local_world <= 1 or not cpus:
        return cpus
    if gpu_numa_node is not None and node_cpus and gpu_numa_node in node_cpus:
        mine = [c for c in sorted(node_cpus[gpu_numa_node]) if c in set(cpus)]
        if mine:
            # ranks are dealt to nodes in order: the ranks sharing this node split its CPUs by their order among themselves
            per_node = max(1, local_world // max(1, len(node_cpus)))
            k = local_rank % per_node
            lo, hi = k * len(mine) // per_node, (k + 1) * len(mine) // per_node
            return mine[lo:hi] or mine
    lo, hi = local_rank * len(cpus) // local_world, (local_rank + 1) * len(cpus) // local_world
    return cpus[lo:hi] or cpus


def _sysfs_topology(local_rank: int) -> tuple[int | None, dict[int, list[int]] | None]:
    """(NUMA node of GPU `local_rank`, {node: cpus}) from sysfs, or (None, None) where the box does not say."""
    try:
        nodes: dict[int, list[int]] = {}
        base = "/sys/devices/system/node"
        for d in os.listdir(base):
            if d.startswith("node") and d[4:].isdigit():
                cl: list[int] = []
                for part in open(os.path.join(base, d, "cpulist")).read().strip().split(","):
                    if part:
                        a, _, b = part.partition("-")
                        cl.extend(range(int(a), int(b or a) + 1))
                nodes[int(d[4:])] = cl
        cards = sorted((c for c in os.listdir("/sys/class/drm") if c.startswith("card") and c[4:].isdigit()), key=lambda c: int(c[4:]))
        gpus = [c for c in cards if os.path.exists(f"/sys/class/drm/{c}/device/numa_node")]
        if local_rank < len(gpus):
            n = int(open(f"/sys/class/drm/{gpus[local_rank]}/device/numa_node").read().strip())
            return (n if n >= 0 else None), (nodes or None)
    except (OSError, ValueError):
        pass
    return None, None


def pin_rank_cpus(local_rank: int, local_world: int) -> list[int]:
    """Restrict this process (and every thread it starts from now on) to rank_cpus(...).  Called by bench.py before the engine
    exists when several ranks share the node (MSH_PIN_CPUS=0: leave the affinity alone).  Returns the CPU list in force."""
    if local_world <= 1 or os.environ.get("MSH_PIN_CPUS", "1") == "0":
        return sorted(os.sched_getaffinity(0))
    node, node_cpus = _sysfs_topology(local_rank)
    cpus = rank_cpus(local_rank, local_world, None, node, node_cpus)
    if len(cpus) < 4:   # a share too small for the lanes' host threads (a CPU-starved container): sharing beats pinning
        return sorted(os.sched_getaffinity(0))
    try:
        os.sched_setaffinity(0, cpus)
    except (OSError, ValueError):
        pass
    return sorted(os.sched_getaffinity(0))


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split; the first (n_items % world) ranks take one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_plan(lens: list[int], world: int) -> list[list[int]]:
    """Which clips go to which rank -- the rule of the C++ host layer (moonshine_amd/csrc/transcriber.cpp,
    MoonshineModel::transcribe_batch): sort by length, longest first (stable), deal the sorted list to the ranks in
    snake order.  Every rank gets the same mix of lengths (equal audio within one clip) and its own list stays sorted, so
    its sub-batches hold clips of similar length.  world == 1 keeps the caller's order."""
    n = len(lens)
    if world <= 1:
        return [list(range(n))]
    order = sorted(range(n), key=lambda i: -lens[i])   # stable: ties keep the caller's order
    plan: list[list[int]] = [[] for _ in range(world)]
    for k, i in enumerate(order):
        rnd, pos = divmod(k, world)
        plan[world - 1 - pos if rnd & 1 else pos].append(i)
    return plan


def scatter_clips(clips: list[np.ndarray] | None, world: int, rank: int, device: torch.device):
    """Rank 0 passes the full clip list, the others None.  Every rank returns (padded [n_local, max_len] float32 tensor on
    ``device``, true lengths, plan) where plan = shard_plan of the whole list (gather_tokens needs it to restore the
    caller's order).  One broadcast of the plan, one scatter of lengths, one scatter of samples."""
    if not use_collectives(world):
        assert clips is not None
        lens = [int(c.shape[0]) for c in clips]
        buf = np.zeros((len(clips), max(lens)), np.float32)
        for i, c in enumerate(clips):
            buf[i, : lens[i]] = c
        return torch.from_numpy(buf).to(device), lens, shard_plan(lens, 1)
    meta = [None]
    if rank == 0:
        lens_all = [int(c.shape[0]) for c in clips]
        meta = [(shard_plan(lens_all, world), max(lens_all))]
    dist.broadcast_object_list(meta, src=0)
    plan, max_len = meta[0]
    per = max(len(p) for p in plan)  # equal-sized scatter chunks (padded with empty clips)
    lens_local = torch.zeros(per, dtype=torch.int64, device=device)
    audio_local = torch.zeros((per, max_len), dtype=torch.float32, device=device)
    if rank == 0:
        lens_chunks, audio_chunks = [], []
        for r in range(world):
            lt = torch.zeros(per, dtype=torch.int64)
            at = torch.zeros((per, max_len), dtype=torch.float32)
            for j, i in enumerate(plan[r]):
                lt[j] = lens_all[i]
                at[j, : lens_all[i]] = torch.from_numpy(np.ascontiguousarray(clips[i], dtype=np.float32))
            lens_chunks.append(lt.to(device))
            audio_chunks.append(at.to(device))
        dist.scatter(lens_local, lens_chunks, src=0)
        dist.scatter(audio_local, audio_chunks, src=0)
    else:
        dist.scatter(lens_local, None, src=0)
        dist.scatter(audio_local, None, src=0)
    k = len(plan[rank])
    return audio_local[:k], [int(v) for v in lens_local[:k].tolist()], plan


def gather_tokens(local: list[list[int]], plan: list[list[int]], world: int, rank: int, device: torch.device) -> list[list[int]]:
    """All ranks receive the token lists of every clip, in the caller's clip order (plan from scatter_clips)."""
    if not use_collectives(world):
        return local
    per = max(len(p) for p in plan)
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
    res: list[list[int]] = [[] for _ in range(sum(len(p) for p in plan))]
    for r in range(world):
        for j, i in enumerate(plan[r]):
            n = int(out[r, j, 0])
            res[i] = out[r, j, 1 : 1 + n].tolist()
    return res


def max_over_ranks(value: float, world: int, device: torch.device) -> float:
    if not use_collectives(world):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
