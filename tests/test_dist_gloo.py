"""Data-parallel plumbing on CPU: 2 ranks over gloo.  Rank 0 owns a RAGGED clip list (lengths from 0.06 s to 10 s, as
real utterance batches are); the list is sharded by the rule of the C++ host layer (length-sorted, snake deal:
moonshine_amd.dist.shard_plan == MoonshineModel::transcribe_batch), scattered, every rank "transcribes" its shard in
engine-shaped sub-batches (sorted by length, padded to the sub-batch's longest clip -- a stand-in that is a pure function
of each clip's own samples, so any mis-routing, padding leak or order mix-up shows), ids are gathered, and every rank must
reconstruct exactly ids(1 process) in the caller's order.  The GPU engine itself is exercised by the -m gpu tests; this
covers plan / scatter / gather / timing-reduction used by bench.py at N > 1."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from moonshine_amd import dist as msd


def _fake_tokens(clip: np.ndarray) -> list[int]:
    n = 1 + int(abs(float(clip[:50].sum())) * 7) % 9
    return [1] + [int(abs(float(clip[(k * 131) % clip.shape[0]])) * 1e4) % 32768 for k in range(n)]


def _fake_engine(audio: torch.Tensor, lens: list[int], sub_batch: int = 3) -> list[list[int]]:
    """What the engine does with a shard: sub-batches in the given (length-sorted) order, each padded to its longest clip."""
    out = []
    for lo in range(0, len(lens), sub_batch):
        part = lens[lo:lo + sub_batch]
        width = max(part)
        block = audio[lo:lo + len(part), :width].numpy()
        for j, n in enumerate(part):
            out.append(_fake_tokens(block[j, :n]))
    return out


def _clips(n, max_len=160000):
    rng = np.random.default_rng(5)
    lens = rng.integers(900, max_len, n)
    lens[1] = lens[4]            # a tie: the stable sort keeps the caller's order
    return [rng.standard_normal(int(k)).astype(np.float32) for k in lens]


def _worker(rank, world, port, n_clips, q, max_len=160000):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cpu")
    r, w = msd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    clips = _clips(n_clips, max_len) if rank == 0 else None
    audio, lens, plan = msd.scatter_clips(clips, world, rank, dev)
    assert audio.shape[0] == len(plan[rank]) == len(lens)
    assert lens == sorted(lens, reverse=True)        # a rank's shard arrives longest first
    local = _fake_engine(audio, lens)
    allt = msd.gather_tokens(local, plan, world, rank, dev)
    tmax = msd.max_over_ranks(1.0 + rank, world, dev)
    q.put((rank, allt, tmax, sum(lens)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _one_rank_worker(port, n_clips, q):
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MSH_DIST_FORCE_GROUP="1")
    dev = torch.device("cpu")
    r, w = msd.init_from_env("gloo")
    assert (r, w) == (0, 1) and torch.distributed.is_initialized() and msd.use_collectives(w)
    clips = _clips(n_clips)
    audio, lens, plan = msd.scatter_clips(clips, w, r, dev)        # the collective path, not the early return
    assert plan == [list(range(n_clips))] and lens == [len(c) for c in clips]   # one rank: the caller's order
    allt = msd.gather_tokens(_fake_engine(audio, lens), plan, w, r, dev)
    q.put((allt, msd.max_over_ranks(3.5, w, dev)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_forced_group_at_one_rank_takes_the_collective_path():
    """MSH_DIST_FORCE_GROUP=1 (moonshine_amd.dist.force_group): a one-rank run creates the group and goes through
    broadcast / scatter / all-reduce / all-gather instead of returning early -- what tests/test_gpu_dist.py runs over
    RCCL on the one-GPU box; here over gloo."""
    assert not msd.use_collectives(1)                                 # default: no group, no collective at world == 1
    n_clips = 7
    want = [_fake_tokens(c) for c in _clips(n_clips)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), n_clips, q))
    p.start()
    allt, tmax = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and allt == want and tmax == 3.5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_ids_equal_single_process():
    n_clips, world = 11, 2  # odd count: uneven shards
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
    for rank, allt, tmax, _ in results:
        assert allt == want, f"rank {rank} reconstructed a different transcript list"     # ids(N) == ids(1)
        assert tmax == 2.0
    audio = sorted(r[3] for r in results)
    assert audio[1] - audio[0] <= 160000              # the shards hold the same amount of audio within one clip


def test_eight_ranks_2048_clips_ids_equal_single_process():
    """BASELINE.json configs[3] in shape -- 2048 ragged clips over 8 ranks (256 per GPU) -- on CPU over gloo: the plan deals
    every rank 256 clips holding the same amount of audio within one clip, every rank reconstructs ids(1 process) in the
    caller's order, and the timing reduction is the maximum over the eight ranks.  (Clips are kept short: the plumbing, not
    the audio, is what eight CPU processes can exercise.)"""
    n_clips, world, max_len = 2048, 8, 6000
    want = [_fake_tokens(c) for c in _clips(n_clips, max_len)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q, max_len)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, allt, tmax, _ in results:
        assert allt == want, f"rank {rank} reconstructed a different transcript list"
        assert tmax == 8.0
    audio = sorted(r[3] for r in results)
    assert audio[-1] - audio[0] <= max_len            # balanced within one clip
    assert sorted(r[0] for r in results) == list(range(world))


def test_shard_plan_properties():
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 256, 2048, 2049):
        lens = rng.integers(1000, 160000, n).tolist()
        for world in (1, 2, 3, 8):
            plan = msd.shard_plan(lens, world)
            assert len(plan) == world and sorted(i for p in plan for i in p) == list(range(n))
            sizes = [len(p) for p in plan]
            assert max(sizes) - min(sizes) <= 1
            if world > 1 and n:
                for p in plan:
                    assert [lens[i] for i in p] == sorted((lens[i] for i in p), reverse=True)
                tot = [sum(lens[i] for i in p) for p in plan]
                assert max(tot) - min(tot) <= max(lens)   # balanced within one clip
    assert msd.shard_plan([5, 9, 7], 1) == [[0, 1, 2]]       # one rank: the caller's order
    assert msd.shard_plan([5, 9, 7, 9], 2) == [[1, 0], [3, 2]]   # ties keep the caller's order; snake: 9a | 9b, 7 | 5


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [msd.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_2048_clip_plan_gives_eight_ranks_the_same_audio():
    """BASELINE config 4 (2,048 clips utterance-sharded over 8 GPUs): the snake deal of the length-sorted list gives every rank
    256 clips whose total audio differs by less than ONE clip between any two ranks -- on equal clips and on a realistic mix of
    lengths (VAD segments: many short, a third at the 10 s cap)."""
    rng = np.random.default_rng(8)
    mixes = [[160000] * 2048,
             (rng.gamma(2.0, 30000.0, 2048).clip(8000, 160000)).astype(int).tolist(),
             [160000] * 700 + rng.integers(8000, 159000, 1348).tolist()]
    for lens in mixes:
        plan = msd.shard_plan(lens, 8)
        assert [len(p) for p in plan] == [256] * 8
        tot = [sum(lens[i] for i in p) for p in plan]
        assert max(tot) - min(tot) < max(lens), (max(tot) - min(tot), max(lens))
        assert (max(tot) - min(tot)) / (sum(tot) / 8) < 1e-2      # ... which is under 1 % of a rank's share


def test_rank_cpu_shares_partition_the_node():
    """moonshine_amd/dist.py rank_cpus: the ranks of a node keep their host threads on disjoint CPU shares that cover the allowed
    set; with topology information a rank stays on its GPU's NUMA node."""
    allowed = list(range(0, 96)) + list(range(128, 224))     # a cgroup-restricted 192-CPU view of a 256-CPU box
    for world in (1, 2, 4, 8):
        shares = [msd.rank_cpus(r, world, allowed) for r in range(world)]
        assert sorted(c for s in shares for c in s) == sorted(allowed)
        assert all(len(s) >= len(allowed) // world for s in shares)
    nodes = {0: list(range(0, 128)), 1: list(range(128, 256))}
    for r in range(8):
        got = msd.rank_cpus(r, 8, allowed, gpu_numa_node=r // 4, node_cpus=nodes)
        assert got and set(got) <= set(nodes[r // 4]) & set(allowed)
    on0 = [msd.rank_cpus(r, 8, allowed, 0, nodes) for r in range(4)]
    assert sorted(c for s in on0 for c in s) == [c for c in allowed if c < 128]     # the four ranks of a node split it
    assert msd.rank_cpus(3, 8, [5]) == [5]                                            # fewer CPUs than ranks: shared


def test_a_cpu_share_too_small_for_the_lanes_is_not_pinned():
    """pin_rank_cpus: a rank whose share of the allowed CPUs is under four (a CPU-starved container: 8 ranks on 16 CPUs) keeps
    the process's affinity -- four lanes' host threads on one or two CPUs would cost more than crossing a socket -- and a
    single rank is never pinned."""
    import os

    before = sorted(os.sched_getaffinity(0))
    world = max(2, len(before))            # at most one CPU per rank
    assert msd.pin_rank_cpus(0, world) == before
    assert sorted(os.sched_getaffinity(0)) == before
    assert msd.pin_rank_cpus(0, 1) == before
