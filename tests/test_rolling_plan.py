"""The plan of the batch call's rolling batch (moonshine_amd/csrc/rolling_plan.{h,cpp}, MoonshineModel::rolling_add) on the host
alone, through msh_host_rolling_plan: which clips go to the GPU after which piece.  No reference counterpart (the reference
transcribes clip after clip, core/transcriber.cpp:997); what is checked is the plan's own contract:

  * every clip goes out exactly once, whatever the pieces;
  * every sub-batch respects run_shard's caps (batch_clips x 10 s of audio once it holds batch_clips clips, 4 x batch_clips clips);
  * a sub-batch submitted before the last piece is FULL and either of nearly one length (shortest >= 0.9 x longest) or made of
    the short class; the one exception is the very first submission (whatever short clips there are, so that the GPU starts);
  * handed over in ONE piece the plan is run_shard's sorted cut: sub-batch k holds the k-th run of the clips sorted longest first;
  * the sum over sub-batches of their longest clip (what the decode costs) stays within 25 % of that sorted cut's for the
    segment-length mix of the benchmark, while at least a fifth of the audio is submitted before the last piece;
  * the plan is a function of the lengths and pieces alone (two runs agree), and narrow_runs = 0 / short_frac = 0 switch the two
    early kinds off.
"""
import ctypes as C

import numpy as np
import pytest

from moonshine_amd.hip_api import load_library


def plan(lens, pieces, batch_clips, short_frac=0.15, narrow=1):
    lib = load_library()
    f = lib.msh_host_rolling_plan
    f.restype = C.c_int64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    lens = np.ascontiguousarray(lens, np.uint64)
    pieces = np.ascontiguousarray(pieces, np.uint64)
    assert int(pieces.sum()) == len(lens)
    sub_of = np.full(len(lens), -1, np.int32)
    cap = len(lens) + 4
    piece_of = np.full(cap, -1, np.int32)
    first_of = np.full(cap, -1, np.int32)
    n = f(lens.ctypes.data, pieces.ctypes.data, len(pieces), batch_clips, short_frac, narrow, sub_of.ctypes.data, piece_of.ctypes.data,
          first_of.ctypes.data, cap)
    assert n >= 0, n
    return int(n), sub_of, piece_of[:n], first_of[:n]


def segment_mix(rng, clips):
    """lengths like the benchmark's segments: a third of the clips unsplit (~10 s), the rest split into two to four pieces"""
    out = []
    for _ in range(clips):
        if rng.random() < 0.35:
            out.append(int(rng.uniform(9.5, 10.0) * 16000))
        else:
            cuts = np.sort(rng.uniform(0.5, 9.5, rng.integers(1, 4)))
            edges = np.concatenate([[0.0], cuts, [10.0]])
            out += [max(int((b - a) * 0.8 * 16000), 8192) for a, b in zip(edges[:-1], edges[1:])]
    return np.asarray(out, np.uint64)


def check_contract(lens, pieces, bc, n, sub_of, piece_of, first_of, short_frac, narrow):
    assert (sub_of >= 0).all() and sub_of.max() == n - 1                      # every clip exactly once
    audio_cap, clip_cap = bc * 160000, min(4 * bc, 1024)
    last_piece = len(pieces) - 1
    order = np.cumsum(pieces)
    piece_of_clip = np.searchsorted(order, np.arange(len(lens)), side="right")
    first = True
    for s in range(n):
        ids = np.nonzero(sub_of == s)[0]
        ln = lens[ids].astype(np.int64)
        assert len(ids) <= clip_cap
        assert (piece_of_clip[ids] <= piece_of[s]).all()                      # nothing goes out before it arrived
        assert int(lens[first_of[s]]) == int(ln.max())
        if len(ids) > bc:                                                     # beyond batch_clips clips only inside the audio budget
            assert int(ln.sum()) <= audio_cap
        if piece_of[s] < last_piece and not first:
            assert ln.sum() * 10 >= audio_cap * 9 or len(ids) == clip_cap or ln.sum() >= audio_cap - ln.max(), (s, ln.sum())
            narrow_run = ln.min() * 10 >= ln.max() * 9
            assert narrow_run or short_frac > 0, s
            if not narrow:
                assert short_frac > 0
        first = False


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_contract_on_the_benchmark_mix(seed):
    rng = np.random.default_rng(seed)
    lens = segment_mix(rng, 2048)
    per_piece = len(lens) // 10
    pieces = [per_piece] * 9 + [len(lens) - 9 * per_piece]
    bc = 256
    n, sub_of, piece_of, first_of = plan(lens, pieces, bc)
    check_contract(lens, pieces, bc, n, sub_of, piece_of, first_of, 0.15, 1)
    again = plan(lens, pieces, bc)
    assert again[0] == n and np.array_equal(again[1], sub_of)
    # cost against the sorted cut of the whole call, and how much goes out early
    n1, sub1, _, first1 = plan(lens, [len(lens)], bc)
    cost = lambda first: int(lens[first].sum())
    assert cost(first_of) <= 1.25 * cost(first1), (cost(first_of), cost(first1))
    early = sum(int(lens[sub_of == s].sum()) for s in range(n) if piece_of[s] < len(pieces) - 1)
    assert early >= 0.2 * int(lens.sum()), early / lens.sum()
    print(f"\n[seed {seed}] {len(lens)} segments: {n} sub-batches ({n1} in one piece), decode cost {cost(first_of) / cost(first1):.3f} x the sorted cut, "
          f"{early / lens.sum():.2f} of the audio submitted before the last piece")


def test_one_piece_is_the_sorted_cut():
    rng = np.random.default_rng(5)
    lens = rng.integers(8192, 160000, 1500).astype(np.uint64)
    bc = 64
    n, sub_of, piece_of, first_of = plan(lens, [len(lens)], bc)
    order = np.argsort(-lens.astype(np.int64), kind="stable")
    assert (np.diff(sub_of[order]) >= 0).all()                                # sub-batches tile the sorted list
    assert (piece_of == 0).all()
    pos = 0
    for s in range(n):                                                        # and each is run_shard's cut at its position
        m, total = 0, 0
        while pos + m < len(order) and m < min(4 * bc, 1024):
            if m >= bc and total + int(lens[order[pos + m]]) > bc * 160000:
                break
            total += int(lens[order[pos + m]])
            m += 1
        assert int((sub_of == s).sum()) == m, s
        pos += m
    assert pos == len(lens)


def test_switches_and_edge_cases():
    rng = np.random.default_rng(9)
    lens = segment_mix(rng, 600)
    pieces = [100] * (len(lens) // 100) + ([len(lens) % 100] if len(lens) % 100 else [])
    bc = 32
    # both early kinds off: only the first submission (the smallest clip class) precedes the last piece
    n, sub_of, piece_of, first_of = plan(lens, pieces, bc, short_frac=0.0, narrow=0)
    assert (piece_of[1:] == len(pieces) - 1).all()
    check_contract(lens, pieces, bc, n, sub_of, piece_of, first_of, 0.0, 0)
    # narrow runs only
    n2, sub2, piece2, first2 = plan(lens, pieces, bc, short_frac=0.0, narrow=1)
    for s in range(1, n2):
        if piece2[s] < len(pieces) - 1:
            ln = lens[sub2 == s].astype(np.int64)
            assert ln.min() * 10 >= ln.max() * 9
    # equal lengths (a call without VAD): every piece of batch_clips clips goes out at once, as one sub-batch
    same = np.full(8 * bc, 160000, np.uint64)
    n3, sub3, piece3, _ = plan(same, [bc] * 8, bc)
    assert n3 == 8 and list(piece3) == list(range(8))
    assert all(((sub3 == s).sum() == bc) for s in range(8))
    # empty pieces, one clip, nothing at all
    assert plan(np.asarray([20000], np.uint64), [0, 1, 0], bc)[0] == 1
    assert plan(np.zeros(0, np.uint64), [], bc)[0] == 0
    assert plan(np.zeros(0, np.uint64), [0, 0], bc)[0] == 0
