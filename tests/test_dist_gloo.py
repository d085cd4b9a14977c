"""Data-parallel plumbing on CPU: 2 ranks over gloo.  Rank 0 scatters a ragged clip list, every rank
"transcribes" its shard with a deterministic stand-in (a pure function of the samples, so any
mis-routing shows), ids are gathered, and every rank must reconstruct exactly what a single process
produces.  The GPU engine itself is exercised by the -m gpu tests; this covers shard / scatter /
gather / timing-reduction used by bench.py at N > 1."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from moonshine_amd import dist as msd


def _fake_tokens(clip: np.ndarray) -> list[int]:
    n = 1 + int(abs(float(clip[:50].sum())) * 7) % 9
    return [1] + [int(abs(float(clip[(k * 131) % clip.shape[0]])) * 1e4) % 32768 for k in range(n)]


def _clips(n):
    rng = np.random.default_rng(5)
    return [rng.standard_normal(int(rng.integers(900, 5000))).astype(np.float32) for _ in range(n)]


def _worker(rank, world, port, n_clips, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cpu")
    r, w = msd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    clips = _clips(n_clips) if rank == 0 else None
    audio, lens = msd.scatter_clips(clips, world, rank, dev)
    lo, hi = msd.shard_bounds(n_clips, rank, world)
    assert audio.shape[0] == hi - lo == len(lens)
    local = [_fake_tokens(audio[i, : lens[i]].numpy()) for i in range(len(lens))]
    allt = msd.gather_tokens(local, n_clips, world, rank, dev)
    tmax = msd.max_over_ranks(1.0 + rank, world, dev)
    q.put((rank, allt, tmax))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_scatter_gather_equals_single_process():
    n_clips, world = 7, 2  # odd count: uneven shards
    want = [_fake_tokens(c) for c in _clips(n_clips)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, allt, tmax in results:
        assert allt == want, f"rank {rank} reconstructed a different transcript list"
        assert tmax == 2.0


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [msd.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
