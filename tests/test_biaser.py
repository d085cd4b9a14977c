"""Known-answer tests of the reference (core/context-biaser-test.cpp, core/bin-tokenizer/bin-tokenizer-test.cpp)
restated against the oracle, and -- through the C ABI helpers -- against the C++ host code of the MI355X build.
CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import biaser_ref as br

V = 64


def bpe_vocab():
    """bin-tokenizer-test.cpp:17-45: control tokens, the 256 raw bytes, then learned pieces."""
    v = [b"<unk>", b"<s>", b"</s>"] + [bytes([i]) for i in range(256)]
    return v


def test_longest_match_known_answers():
    vocab = [b"", b"a", b"ab", b"b", b"abc", b"ab"]                     # bin-tokenizer-test.cpp:73-91
    f = lambda t: br.text_to_tokens_longest_match(vocab, t, b"_")
    assert f(b"abc") == [4] and f(b"ab") == [2] and f(b"aba") == [2, 1] and f(b"ba") == [3, 1]
    with pytest.raises(ValueError):
        f(b"z")


def test_bpe_known_answers():
    vocab = bpe_vocab()                                                   # bin-tokenizer-test.cpp:93-143
    cd, ab, abc = len(vocab), len(vocab) + 1, len(vocab) + 2
    vocab += [b"cd", b"ab", b"abc"]
    bid = lambda c: 3 + (c if isinstance(c, int) else ord(c))
    f = lambda t: br.text_to_tokens_bpe(vocab, t, b"_")
    assert f(b"abcd") == [ab, cd]
    assert br.text_to_tokens_longest_match(vocab, b"abcd", b"_") == [abc, bid("d")]
    assert f(b"abc") == [abc] and f(b"ab") == [ab] and f(b"cd") == [cd]
    assert f(b"z") == [bid("z")]
    assert f("é".encode()) == [bid(0xC3), bid(0xA9)]
    assert f(b" ab") == [bid("_"), ab]
    # no byte block: falls back to longest match (bin-tokenizer-test.cpp:144-150)
    assert br.text_to_tokens_bpe([b"", b"a", b"ab"], b"ab", b"_") == [2]


def test_context_biaser_known_answers():
    z = lambda: np.zeros(V, np.float32)
    b = br.ContextBiaser()                                                # empty-biaser-changes-nothing
    lg = z(); b.apply(lg); assert b.empty() and not lg.any()
    b = br.ContextBiaser(0.0); b.add_token_sequence([10, 11])             # zero-boost-adds-nothing
    lg = z(); b.apply(lg); assert not lg.any()
    b = br.ContextBiaser(2.0); b.add_token_sequence([10, 11, 12])         # boosts-only-the-first-token
    lg = z(); b.apply(lg)
    assert lg[10] == pytest.approx(2.0) and lg[11] == 0 and lg[12] == 0
    assert b.bonus_for_token(10) == pytest.approx(2.0)                    # bonus-grows-with-depth
    b.advance(10); assert b.bonus_for_token(11) == pytest.approx(2.0 * (1 + math.log(2.0)))
    b.advance(11); assert b.bonus_for_token(12) == pytest.approx(2.0 * (1 + math.log(3.0)))
    b = br.ContextBiaser(); b.add_token_sequence([10, 11, 12])            # abandoning-a-path
    b.advance(10); assert b.bonus_for_token(11) > 0
    b.advance(42); assert b.bonus_for_token(11) == 0 and b.bonus_for_token(10) > 0
    b = br.ContextBiaser(); b.add_token_sequence([10, 11]); b.add_token_sequence([11, 20])
    b.advance(10); b.advance(11); assert b.bonus_for_token(20) > 0        # a-term-can-start-part-way
    b = br.ContextBiaser(3.0); b.add_token_sequence([10, 11]); b.add_token_sequence([10, 12])
    lg = z(); b.apply(lg); assert lg[10] == pytest.approx(3.0)            # shared-prefixes-are-not-stacked
    b = br.ContextBiaser(3.0); b.add_token_sequence([10, 11]); b.add_token_sequence([11, 20])
    b.advance(10); lg = z(); b.apply(lg)                                  # takes-the-larger-bonus
    assert lg[11] == pytest.approx(3.0 * (1 + math.log(2.0))) and lg[10] == pytest.approx(3.0)
    b = br.ContextBiaser(); b.add_token_sequence([10, 11]); b.advance(10); b.reset()
    assert b.bonus_for_token(11) == 0 and b.bonus_for_token(10) > 0       # reset-discards-a-partial-match
    b = br.ContextBiaser(); b.add_token_sequence([V + 5, 3]); lg = z(); b.apply(lg); assert not lg.any()
    b = br.ContextBiaser(); b.add_token_sequence([]); assert b.empty()
    assert br.ContextBiaser.variants_for_term("Kubernetes") == ["Kubernetes", " Kubernetes"]
    assert br.ContextBiaser.variants_for_term("  Kubernetes  ")[0] == "Kubernetes"
    assert br.ContextBiaser.variants_for_term("   ") == []
    assert br.ContextBiaser.variants_for_term("▁Ku") == ["▁Ku"]


# ---- the C++ host code of this build, through the C ABI test helpers ----
@pytest.fixture(scope="module")
def lib():
    from moonshine_amd.hip_api import load_library

    l = load_library()
    l.msh_host_text_to_tokens.restype = C.c_int64
    l.msh_host_text_to_tokens.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_int32,
                                          C.c_void_p, C.c_uint64]
    l.msh_host_biaser_bonuses.restype = C.c_int64
    l.msh_host_biaser_bonuses.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p, C.c_uint64,
                                          C.c_void_p, C.c_uint64]
    return l


def cxx_encode(lib, vocab, text: bytes, space: bytes, bpe: bool):
    from moonshine_amd.synth import encode_tokenizer_bin

    blob = encode_tokenizer_bin(vocab)
    out = np.zeros(256, np.int32)
    n = lib.msh_host_text_to_tokens(blob, len(blob), text, len(text), space, 1 if bpe else 0, out.ctypes.data, out.shape[0])
    if n < 0:
        raise ValueError("encode failed")
    return out[:n].tolist()


def test_cxx_tokenizer_encode_matches_oracle(lib):
    vocab = bpe_vocab() + [b"cd", b"ab", b"abc", SPACE_PIECE, b"\xe2\x96\x81ab", b"xyz", b"ab"]
    rng = np.random.default_rng(4)
    texts = [b"abcd", b"abc", b" ab", b"z", "é".encode(), b"ab ab cd", b"xyzab", b""]
    alphabet = b"abcdxyz \xc3\xa9"
    texts += [bytes(rng.choice(list(alphabet), size=int(rng.integers(1, 12))).tolist()) for _ in range(60)]
    for t in texts:
        assert cxx_encode(lib, vocab, t, br.SPACE, True) == br.text_to_tokens_bpe(vocab, t)
    lm = [b"", b"a", b"ab", b"b", b"abc", b"ab"]
    for t in (b"abc", b"ab", b"aba", b"ba"):
        assert cxx_encode(lib, lm, t, b"_", False) == br.text_to_tokens_longest_match(lm, t, b"_")
        assert cxx_encode(lib, lm, t, b"_", True) == br.text_to_tokens_longest_match(lm, t, b"_")   # no byte block
    with pytest.raises(ValueError):
        cxx_encode(lib, lm, b"z", b"_", False)


SPACE_PIECE = br.SPACE


def cxx_bonuses(lib, seqs, boost, prefix):
    flat = np.asarray([t for s in seqs for t in s], np.int32)
    lens = np.asarray([len(s) for s in seqs], np.int32)
    pre = np.asarray(prefix, np.int32)
    out = np.zeros(V, np.float32)
    n = lib.msh_host_biaser_bonuses(flat.ctypes.data, lens.ctypes.data, len(seqs), boost, pre.ctypes.data, len(prefix),
                                    out.ctypes.data, V)
    assert n >= 0
    return out


def test_cxx_biaser_matches_oracle(lib):
    rng = np.random.default_rng(9)
    for trial in range(40):
        seqs = [rng.integers(3, 20, size=int(rng.integers(1, 5))).tolist() for _ in range(int(rng.integers(1, 8)))]
        boost = float(rng.choice([2.0, 3.0, 0.5]))
        prefix = rng.integers(3, 20, size=int(rng.integers(0, 10))).tolist()
        b = br.ContextBiaser(boost)
        for s in seqs:
            b.add_token_sequence(s)
        for t in prefix:
            b.advance(t)
        want = np.zeros(V, np.float32)
        b.apply(want)
        np.testing.assert_allclose(cxx_bonuses(lib, seqs, boost, prefix), want, rtol=1e-6, atol=0)


# ---- ContextExtractor: the reference's known-answer cases (core/context-extractor-test.cpp) ----
COMMON = {"the", "and", "for", "with", "about", "some", "will", "have", "team", "meeting", "notes", "word", "here",
          "chapter", "doctor", "prison", "wine", "shop", "spy", "said", "very"}


def stub_subword_count(word: bytes) -> int:
    bare = word.strip(b" \t").lower()
    if not bare:
        return 0
    if bare.decode("utf-8", "replace") in COMMON:
        return 1
    return (len(bare) + 2) // 3


def test_context_extractor_words_known_answers():
    cw = lambda t: [w.decode() for w in br.candidate_words(t)]
    assert cw("The meeting, with Defarge: ready?") == ["The", "meeting", "with", "Defarge", "ready"]
    assert cw("a wine-shop, and Marie's can't") == ["wine-shop", "and", "Marie", "can't"]
    assert cw("Defarge -- Manette") == ["Defarge", "Manette"]
    assert cw("Tellson’s bank—Lorry waited") == ["Tellson", "bank", "Lorry", "waited"]
    assert cw("an IPv6 route to 10 hosts in Kubernetes") == ["route", "hosts", "Kubernetes"]
    w = cw("the Marquis St. Évrémonde")
    assert "Marquis" in w and "Évrémonde" in w
    assert cw("два дв") == ["два"]
    assert br.strip_possessive(b"Tellson's") == b"Tellson" and br.strip_possessive(b"Jones'") == b"Jones"
    assert br.strip_possessive(b"Defarge") == b"Defarge"
    assert cw("Tellson's clerk and Jones' desk") == ["Tellson", "clerk", "and", "Jones", "desk"]


def test_context_extractor_extract_known_answers():
    ex = lambda t, m=0, f=stub_subword_count: [w.decode() for w in br.extract_terms(t, m, f)]
    terms = ex("The team meeting notes said very little about Kubernetes.")
    assert "Kubernetes" in terms and not {"meeting", "notes", "The"} & set(terms)
    terms = ex("Defarge and Defarge and Defarge, with Manette.")
    assert terms == ["Defarge", "Manette"]
    terms = ex("Ceph and glomerulonephritis.")
    assert terms.index("glomerulonephritis") < terms.index("Ceph")
    terms = ex("Madame Defarge, madame Defarge, Madame Defarge, and MADAME.")
    assert "Madame" in terms and "madame" not in terms and "MADAME" not in terms
    assert terms.index("Madame") < terms.index("Defarge")
    assert ex("Defarge Defarge Defarge Manette Manette Cruncher", 2) == ["Defarge", "Manette"]
    many = "".join(f"zq{chr(a)}{chr(b)}vx " for a in range(97, 123) for b in range(97, 123))
    assert len(ex(many, 0)) == br.DEFAULT_MAX_TERMS == len(ex(many, -1))
    assert ex("") == [] and ex("   ,,,   ") == []
    assert ex("Defarge and Manette", 0, lambda w: 0) == []


def test_cxx_context_extractor_matches_oracle(lib):
    from moonshine_amd.synth import encode_tokenizer_bin

    lib.msh_host_context_terms.restype = C.c_int64
    lib.msh_host_context_terms.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_uint64]
    # a BPE vocabulary: raw bytes, the marker, and a few merges so that common words are one token
    vocab = bpe_vocab()
    for piece in ["th", "the", "an", "and", "in", "er", "on", "at", "ng", "ing", "re", "De", "Def", "Ma", "Man"]:
        vocab.append(piece.encode())
    for piece in ["the", "and", "team", "with", "said"]:
        vocab.append(br.SPACE + piece.encode())
    vocab.append(br.SPACE)
    blob = encode_tokenizer_bin(vocab)
    count = lambda w: len(br.text_to_tokens_bpe(vocab, w))
    passages = [
        "The team meeting notes said very little about Kubernetes.",
        "Defarge and Defarge and Defarge, with Manette. Madame Defarge, madame Defarge — Tellson’s bank.",
        "Ceph and glomerulonephritis; an IPv6 route to 10 hosts, the wine-shop and Jones' desk, Évrémonde!",
        "", "   ,,,   ", "the and the and with",
    ]
    for text in passages:
        for cap in (0, 3):
            want = [w.decode() for w in br.extract_terms(text, cap, count)]
            raw = text.encode()
            out = C.create_string_buffer(65536)
            n = lib.msh_host_context_terms(blob, len(blob), raw, len(raw), cap, out, 65536)
            assert n >= 0
            got = out.raw[:n].decode().split("\n") if n else []
            assert got == want, (text, cap, got, want)
