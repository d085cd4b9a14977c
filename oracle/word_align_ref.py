"""Word timestamps from decoder cross-attention: CPU restatement (numpy / plain loops) of the reference's
core/word-alignment.cpp.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  dtw            word-alignment.cpp:12-88    cumulative cost, ties -> diagonal, then "text retreats", then "time retreats"
  median_filter  word-alignment.cpp:98-153   odd width, reflect padding WITHOUT edge repeat (numpy mode "reflect"),
                                             out-of-range reflections clamped for rows narrower than the pad
  align_words    word-alignment.cpp:181-394  z-score over frames per (head, step) -> median 7 -> mean over heads ->
                                             DTW on the negated matrix -> group tokens into words at U+2581 -> frame
                                             span per word -> overlaps snapped to the midpoint
Pinned by: scipy.ndimage.median_filter(mode="mirror") and a brute-force minimum-cost monotone path search for the two
numeric kernels (tests/test_word_alignment.py); the reference's own test (word-alignment-test.cpp) needs the shipped
tiny-en model and only states properties (end > start, monotone starts, confidence in [0, 1]), which the GPU tests
assert on the product's output.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
WORD_MARK = b"\xe2\x96\x81"  # U+2581, the SentencePiece word-boundary marker


def dtw(cost: np.ndarray) -> tuple[list[int], list[int]]:
    """cost [N, M] -> (text indices, time indices) of the cheapest monotone path from (0,0) to (N-1,M-1)."""
    cost = np.asarray(cost, F32)
    N, M = cost.shape
    D = np.full((N + 1, M + 1), np.inf, F32)
    D[0, 0] = 0
    trace = np.zeros((N, M), np.int8)
    for i in range(N):
        for j in range(M):
            c0, c1, c2 = D[i, j], D[i, j + 1], D[i + 1, j]
            if c0 <= c1 and c0 <= c2:
                t, m = 0, c0
            elif c1 <= c0 and c1 <= c2:
                t, m = 1, c1
            else:
                t, m = 2, c2
            trace[i, j] = t
            D[i + 1, j + 1] = F32(cost[i, j] + m)
    i, j = N - 1, M - 1
    ti, tj = [], []
    while i >= 0 or j >= 0:
        ti.append(i)
        tj.append(j)
        if i == 0 and j == 0:
            break
        t = trace[i, j]
        if t == 0:
            i, j = i - 1, j - 1
        elif t == 1:
            i -= 1
        else:
            j -= 1
    return ti[::-1], tj[::-1]


def median_filter(x: np.ndarray, width: int) -> np.ndarray:
    """Median along the last axis; even widths are bumped to the next odd one."""
    x = np.asarray(x, F32)
    if width <= 1:
        return x.copy()
    if width % 2 == 0:
        width += 1
    pad = width // 2
    W = x.shape[-1]
    left = [min(pad - p, W - 1) for p in range(pad)]
    right = [max(W - 2 - p, 0) for p in range(pad)]
    idx = np.asarray(left + list(range(W)) + right)
    padded = x[..., idx]
    win = np.stack([padded[..., k:k + W] for k in range(width)], axis=-1)
    return np.sort(win, axis=-1)[..., width // 2].astype(F32)  # nth_element(n/2) of an odd window = the median


def _starts_word(vocab: list[bytes], tok: int) -> bool:
    return 0 <= tok < len(vocab) and vocab[tok][:3] == WORD_MARK


def align_words(att: np.ndarray, tokens: list[int], time_per_frame: float, vocab: list[bytes], tokens_to_text) -> list[dict]:
    """att [L*H, steps, frames] fp32; tokens = [BOS, t1, ..., tN, EOS-or-last]; tokens_to_text(vocab, ids) -> bytes is
    the tokenizer's detokeniser.  Returns [{text, start, end, confidence}]."""
    att = np.asarray(att, F32)
    if att.size == 0:
        return []
    heads, steps, frames = att.shape
    mean = (att.sum(axis=-1, dtype=F32) / F32(frames))[..., None]
    std = np.sqrt(((att - mean) ** 2).sum(axis=-1, dtype=F32) / F32(frames))[..., None]
    std = np.where(std == 0, F32(1e-10), std)
    w = median_filter(((att - mean) / std).astype(F32), 7)
    matrix = (w.sum(axis=0, dtype=F32) * F32(1.0 / heads)).astype(F32)
    ti, tj = dtw(-matrix)
    text = list(tokens[1:-1]) if len(tokens) >= 2 else []
    if not text:
        return []
    groups: list[tuple[list[int], list[int]]] = []
    cur_t: list[int] = []
    cur_s: list[int] = []
    for i, tok in enumerate(text):
        if _starts_word(vocab, tok) and cur_t:
            groups.append((cur_t, cur_s))
            cur_t, cur_s = [], []
        cur_t.append(tok)
        cur_s.append(i)
    if cur_t:
        groups.append((cur_t, cur_s))
    out = []
    for toks, rows in groups:
        txt = tokens_to_text(vocab, toks).strip(b" \t\n\r")
        if not txt:
            continue
        fr = [tj[p] for p in range(len(ti)) if ti[p] in rows]
        if fr:
            start, end = F32(min(fr)) * F32(time_per_frame), F32(max(fr) + 1) * F32(time_per_frame)
        else:
            start = end = F32(0)
        out.append({"text": txt, "start": F32(start), "end": F32(end), "confidence": F32(1.0)})
    for a, b in zip(out, out[1:]):
        if a["end"] > b["start"]:
            mid = F32((a["end"] + b["start"]) * F32(0.5))
            a["end"] = mid
            b["start"] = mid
    return out
