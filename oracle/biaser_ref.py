"""Restatement of the reference's contextual biasing and of the tokenizer's
text -> ids direction it depends on.

TEST INFRASTRUCTURE (see oracle/__init__.py).

  ContextBiaser   reference core/context-biaser.{h,cpp} (cited ``cb:<line>``)
  text_to_tokens  reference core/bin-tokenizer/bin-tokenizer.cpp:234-402 (cited ``bt:<line>``)

Pinned on the known-answer cases of the reference's own tests, core/context-biaser-test.cpp and
core/bin-tokenizer/bin-tokenizer-test.cpp (restated in tests/test_biaser.py).
"""
from __future__ import annotations

import math

import numpy as np

SPACE = "▁".encode("utf-8")


# --- tokenizer: text -> ids ------------------------------------------------------------------------
def _utf8_len(lead: int) -> int:
    """bt:18-32."""
    if lead & 0x80 == 0x00:
        return 1
    if lead & 0xE0 == 0xC0:
        return 2
    if lead & 0xF0 == 0xE0:
        return 3
    if lead & 0xF8 == 0xF0:
        return 4
    return 1


def byte_fallback_base(vocab: list[bytes]) -> int:
    """bt:238-254: first run of 256 consecutive single-byte entries 0x00..0xFF, or -1."""
    n = len(vocab)
    for start in range(0, n - 255):
        if all(len(vocab[start + o]) == 1 and vocab[start + o][0] == o for o in range(256)):
            return start
    return -1


def text_to_tokens_longest_match(vocab: list[bytes], text: bytes, space: bytes = SPACE) -> list[int]:
    """bt:292-340: greedy longest match, lowest id among equal lengths; raises when nothing matches."""
    rem = text.replace(b" ", space)
    out = []
    while rem:
        best, best_len = -1, 0
        for i, b in enumerate(vocab):
            if b and len(b) > best_len and rem.startswith(b):
                best, best_len = i, len(b)
        if best < 0:
            raise ValueError(f"No match found for remaining bytes {rem!r}")
        out.append(best)
        rem = rem[best_len:]
    return out


def text_to_tokens_bpe(vocab: list[bytes], text: bytes, space: bytes = SPACE) -> list[int]:
    """bt:347-396: ids stand in for merge ranks; pieces after the byte block only; lowest id merges first;
    unspellable characters fall back to their raw bytes.  Falls back to longest match without a byte block
    (bt:255-260)."""
    base = byte_fallback_base(vocab)
    if base < 0:
        return text_to_tokens_longest_match(vocab, text, space)
    merge = {}
    for i in range(base + 256, len(vocab)):
        if vocab[i] and vocab[i] not in merge:
            merge[vocab[i]] = i
    s = text.replace(b" ", space)
    pieces, off = [], 0
    while off < len(s):
        n = min(_utf8_len(s[off]), len(s) - off)
        pieces.append(s[off:off + n])
        off += n
    while len(pieces) > 1:
        best_id, best_pos = -1, 0
        for p in range(len(pieces) - 1):
            cid = merge.get(pieces[p] + pieces[p + 1])
            if cid is not None and (best_id < 0 or cid < best_id):
                best_id, best_pos = cid, p
        if best_id < 0:
            break
        pieces[best_pos:best_pos + 2] = [pieces[best_pos] + pieces[best_pos + 1]]
    out = []
    for piece in pieces:
        if piece in merge:
            out.append(merge[piece])
        else:
            out.extend(base + b for b in piece)
    return out


# --- ContextBiaser -------------------------------------------------------------------------------------
class ContextBiaser:
    DEFAULT_BOOST = 2.0  # cb.h:43

    def __init__(self, boost: float = DEFAULT_BOOST):
        self.boost = np.float32(boost)
        self.children = [dict()]   # node -> {token: child}
        self.depth = [0]
        self.sequence_count = 0
        self.active = [0]

    def add_token_sequence(self, tokens):
        """cb:17-39."""
        if not tokens:
            return
        node = 0
        for t in tokens:
            nxt = self.children[node].get(int(t))
            if nxt is None:
                nxt = len(self.children)
                self.children[node][int(t)] = nxt
                self.children.append(dict())
                self.depth.append(self.depth[node] + 1)
            node = nxt
        self.sequence_count += 1

    @staticmethod
    def variants_for_term(term: str) -> list[str]:
        """cb:41-52: trimmed, plus the word-start form unless the caller already anchored it."""
        t = term.strip(" \t")
        if not t:
            return []
        if t.startswith("▁"):
            return [t]
        return [t, " " + t]

    def empty(self) -> bool:
        return self.sequence_count == 0

    def reset(self):
        self.active = [0]

    def bonus_for_depth(self, depth: int) -> np.float32:
        """cb:63-68: boost * (1 + ln depth) in float32."""
        if depth <= 0:
            return np.float32(0.0)
        return np.float32(self.boost * (np.float32(1.0) + np.float32(math.log(np.float32(depth)))))

    def pending(self, vocab_size: int) -> dict[int, np.float32]:
        """cb:88-131: candidate tokens of every active node, the larger bonus where two nodes agree."""
        out: dict[int, np.float32] = {}
        if self.sequence_count == 0:
            return out
        for node in self.active:
            bonus = self.bonus_for_depth(self.depth[node] + 1)
            for tok in self.children[node]:
                if 0 <= tok < vocab_size:
                    out[tok] = max(out[tok], bonus) if tok in out else bonus
        return out

    def apply(self, logits: np.ndarray):
        for tok, bonus in self.pending(logits.shape[0]).items():
            logits[tok] += bonus

    def advance(self, token: int):
        """cb:134-149: the root stays active."""
        if self.sequence_count == 0:
            return
        nxt = [0]
        for node in self.active:
            c = self.children[node].get(int(token))
            if c is not None:
                nxt.append(c)
        self.active = nxt

    def bonus_for_token(self, token: int) -> np.float32:
        best = np.float32(0.0)
        for node in self.active:
            c = self.children[node].get(int(token))
            if c is not None:
                best = max(best, self.bonus_for_depth(self.depth[c]))
        return best


# --- ContextExtractor: key terms out of a free-text passage ---------------------------------------------
# reference core/context-extractor.{h,cpp} (cited ``ce:<line>``)
_PUNCT = [("’", "'"), ("‘", "'"), ("“", " "), ("”", " "), ("–", " "), ("—", " "),
          ("…", " "), (" ", " ")]
DEFAULT_MAX_TERMS = 200   # ce.h:36
MIN_SUBWORDS = 2          # ce.h:43
MIN_CHARACTERS = 3        # ce.h:48


def strip_possessive(word: bytes) -> bytes:
    """ce:102-110."""
    if word.endswith(b"'s") or word.endswith(b"'S"):
        return word[:-2]
    if word.endswith(b"'"):
        return word[:-1]
    return word


def candidate_words(text: str) -> list[bytes]:
    """ce:112-150: letters (any byte >= 0x80 counts as one) and digits build a word, ' and - only inside one;
    possessives stripped, words under three characters or holding a digit dropped; case kept."""
    for a, b in _PUNCT:
        text = text.replace(a, b)
    data = text.encode("utf-8")
    is_letter = lambda c: c >= 0x80 or (65 <= c <= 90) or (97 <= c <= 122)
    is_digit = lambda c: 48 <= c <= 57
    is_join = lambda c: c in (39, 45)
    words, cur = [], bytearray()

    def flush():
        nonlocal cur
        if not cur:
            return
        w = bytes(cur).strip(b"'-")
        w = strip_possessive(w).strip(b"'-")
        cur = bytearray()
        if sum(1 for c in w if (c & 0xC0) != 0x80) < MIN_CHARACTERS:
            return
        if any(is_digit(c) for c in w):
            return
        words.append(w)

    for c in data:
        if is_letter(c) or is_digit(c):
            cur.append(c)
        elif is_join(c) and cur:
            cur.append(c)
        else:
            flush()
    flush()
    return words


def extract_terms(context: str, max_terms: int, subword_count) -> list[bytes]:
    """ce:152-226: group case variants (ASCII fold), keep the majority spelling (earliest on a tie), require
    >= 2 subwords of " " + term, rank by occurrences, then subwords, then first appearance; cap the list."""
    limit = max_terms if max_terms > 0 else DEFAULT_MAX_TERMS
    words = candidate_words(context)
    forms: dict[bytes, list[int]] = {}
    for i, w in enumerate(words):
        if w not in forms:
            forms[w] = [0, i]
        forms[w][0] += 1
    groups: dict[bytes, dict] = {}
    for form, (count, first) in forms.items():
        key = bytes(c + 32 if 65 <= c <= 90 else c for c in form)
        g = groups.setdefault(key, {"term": None, "term_count": 0, "first": 0, "occ": 0})
        g["occ"] += count
        if g["term"] is None or count > g["term_count"] or (count == g["term_count"] and first < g["first"]):
            g["term"], g["term_count"], g["first"] = form, count, first
    cands = []
    for g in groups.values():
        sub = subword_count(b" " + g["term"])
        if sub >= MIN_SUBWORDS:
            cands.append((-g["occ"], -sub, g["first"], g["term"]))
    cands.sort()
    return [c[3] for c in cands[:limit]]
