"""CPU oracle for the Moonshine encoder-decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / CPU baseline.  The shipped path
(``moonshine_amd`` -> ``libmoonshine.so`` -> HIP kernels) never calls into it
and has no CPU fallback.

Parity pinning: the reference hot path executes inside ONNX Runtime 1.23.2 on
``.ort`` graphs that are not in the checkout, and the
reference's own tests hold no numeric golden vectors for this path
(SURVEY.md section 8c).  The float definition the reference itself names as its
oracle is HuggingFace ``transformers`` ``modeling_moonshine.py`` (reference
``docs/models/accuracy.md:14-19``, ``scripts/eval-librispeech.py:434-477``).
This restatement is pinned against outputs of that implementation generated in
the build container by ``tests/golden/make_golden.py`` and committed under
``tests/golden/`` (see ``tests/test_oracle_golden.py``).

Modules and what pins them:
  moonshine_ref.py   offline encoder / decoder / greedy loop     <- HF MoonshineForConditionalGeneration vectors
  streaming_ref.py   the five streaming graphs + their C++ driver <- outputs of the reference's own graph wrapper
                     (frontend, window encoder, adapter, cross-KV,   modules (lora/export.py) over HF
                     multi-token decoder, speculative decode_full)   MoonshineStreaming, tests/golden/make_golden_streaming.py
  biaser_ref.py      ContextBiaser + tokenizer text -> ids        <- the known-answer cases of the reference's
                                                                     context-biaser-test.cpp / bin-tokenizer-test.cpp
  word_align_ref.py  DTW, median filter, align_words (word timestamps) <- scipy.ndimage.median_filter(mode="mirror") and a brute-force
                                                                     search over all monotone paths (tests/test_word_alignment.py)
  host_ref.py        tokenizer.bin, tokens_to_text, sanitize_text, VAD (threshold 0), resampler, step budgets
                     (byte / integer rules restated from the cited C++; the reference holds no fixtures for them)
Unpinned everywhere: agreement with the shipped int8 ``.ort`` graphs themselves (absent from the checkout).
"""
