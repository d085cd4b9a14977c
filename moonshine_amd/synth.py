"""Synthetic Moonshine weights (HF tensor names), synthetic clips, and a minimal safetensors
reader/writer.  Data generation only -- no model arithmetic lives here.

The reference's shipped weights are int8 ``.ort`` flatbuffers fetched from a CDN
(reference ``core/moonshine-model-catalog.cpp:86-113``) and the float
checkpoints live on the HuggingFace hub; neither is on disk and there is no
network.  Tests and benchmarks therefore run on deterministic synthetic weights
that use the HuggingFace state_dict names and shapes
(``transformers/models/moonshine/modeling_moonshine.py:520-540, 265-274, 74-75,
89-90, 836-850``) so a real ``UsefulSensors/moonshine-base`` checkpoint drops in
unchanged.

Weight distribution "fanin-v1": matrices ~ N(0, gain/sqrt(fan_in)) so that
activations, attention scores and logits are O(1) (a 0.02-std init makes every
softmax nearly uniform and hides masking / scaling bugs); biases ~ N(0, 0.1);
norm scales ~ 1 + 0.1 N(0,1).  Every tensor has its own PCG64 stream keyed by
(seed, tensor index), so generation is order-independent and reproducible on
any machine.
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class ArchConfig:
    """Dimensions of a non-streaming Moonshine model.

    base / tiny follow reference ``core/moonshine-model.cpp:41-54`` and
    ``configuration_moonshine.py:80-112``; ``micro`` is a test-only shrink.
    """

    name: str
    hidden: int          # D
    ffn: int             # F (encoder fc1 out; decoder fc1 out is 2F)
    enc_layers: int
    dec_layers: int
    heads: int
    vocab: int = 32768
    bos: int = 1         # reference core/moonshine-model.cpp:56
    eos: int = 2         # reference core/moonshine-model.cpp:57
    rope_theta: float = 10000.0
    partial_rotary: float = 0.9

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def rotary_dim(self) -> int:
        # modeling_moonshine.py:132-134  dim = int(head_dim * partial_rotary_factor)
        return int(self.head_dim * self.partial_rotary)


ARCHS = {
    "tiny": ArchConfig("tiny", 288, 1152, 6, 6, 8),
    "base": ArchConfig("base", 416, 1664, 8, 8, 8),
    # test-only: small enough that the pure-numpy oracle runs in milliseconds
    "micro": ArchConfig("micro", 64, 256, 2, 2, 4, vocab=512),
}


def tensor_specs(cfg: ArchConfig):
    """(name, shape, kind, fan_in) in HF state_dict order. kind in
    {matrix, bias, scale, conv1}."""
    D, F, V = cfg.hidden, cfg.ffn, cfg.vocab
    specs = [
        ("model.encoder.conv1.weight", (D, 1, 127), "conv1", 127),
        ("model.encoder.conv2.weight", (2 * D, D, 7), "matrix", 7 * D),
        ("model.encoder.conv2.bias", (2 * D,), "bias", 0),
        ("model.encoder.conv3.weight", (D, 2 * D, 3), "matrix", 6 * D),
        ("model.encoder.conv3.bias", (D,), "bias", 0),
        ("model.encoder.groupnorm.weight", (D,), "scale", 0),
        ("model.encoder.groupnorm.bias", (D,), "bias", 0),
    ]
    for l in range(cfg.enc_layers):
        p = f"model.encoder.layers.{l}."
        for n in ("q", "k", "v", "o"):
            specs.append((p + f"self_attn.{n}_proj.weight", (D, D), "matrix", D))
        specs += [
            (p + "mlp.fc1.weight", (F, D), "matrix", D),
            (p + "mlp.fc1.bias", (F,), "bias", 0),
            (p + "mlp.fc2.weight", (D, F), "matrix", F),
            (p + "mlp.fc2.bias", (D,), "bias", 0),
            (p + "input_layernorm.weight", (D,), "scale", 0),
            (p + "post_attention_layernorm.weight", (D,), "scale", 0),
        ]
    specs.append(("model.encoder.layer_norm.weight", (D,), "scale", 0))
    specs.append(("model.decoder.embed_tokens.weight", (V, D), "matrix", D))
    for l in range(cfg.dec_layers):
        p = f"model.decoder.layers.{l}."
        for a in ("self_attn", "encoder_attn"):
            for n in ("q", "k", "v", "o"):
                specs.append((p + f"{a}.{n}_proj.weight", (D, D), "matrix", D))
        specs += [
            (p + "mlp.fc1.weight", (2 * F, D), "matrix", D),
            (p + "mlp.fc1.bias", (2 * F,), "bias", 0),
            (p + "mlp.fc2.weight", (D, F), "matrix", F),
            (p + "mlp.fc2.bias", (D,), "bias", 0),
            (p + "input_layernorm.weight", (D,), "scale", 0),
            (p + "post_attention_layernorm.weight", (D,), "scale", 0),
            (p + "final_layernorm.weight", (D,), "scale", 0),
        ]
    specs.append(("model.decoder.norm.weight", (D,), "scale", 0))
    return specs


def make_weights(cfg: ArchConfig, seed: int = 0) -> dict[str, np.ndarray]:
    """Deterministic synthetic fp32 weights, HF names. ``proj_out.weight`` is
    tied to the embedding (configuration_moonshine.py:103) and is not stored."""
    out: dict[str, np.ndarray] = {}
    for idx, (name, shape, kind, fan_in) in enumerate(tensor_specs(cfg)):
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        x = rng.standard_normal(shape, dtype=np.float32)
        if kind == "matrix":
            x *= np.float32(1.0 / np.sqrt(fan_in))
        elif kind == "conv1":
            x *= np.float32(3.0 / np.sqrt(fan_in))
        elif kind == "bias":
            x *= np.float32(0.1)
        elif kind == "scale":
            x = np.float32(1.0) + np.float32(0.1) * x
        out[name] = np.ascontiguousarray(x, dtype=np.float32)
    return out


SHARP_EMBED_GAIN = 4.0


def sharp_weights(cfg: ArchConfig, seed: int = 0, gain: float = SHARP_EMBED_GAIN) -> dict[str, np.ndarray]:
    """``make_weights`` with the tied embedding (input lookup AND LM head) multiplied by ``gain``: the "sharpened" synthetic
    checkpoint of tests/test_gpu_long_parity.py and bench.py's ``cpu_baseline`` id comparison.  With fan-in-scaled random
    weights the logits are ~N(0, 1) over 32768 entries, so ~20 % of the decode positions have a top-1 margin below the 0.1
    id tolerance and a free-running 65-step comparison almost always meets one (a trained checkpoint is far more peaked).
    The gain puts the token's own embedding in charge of the residual stream: >= 95 % of the positions clear 0.1 and
    free-running ids become comparable clip by clip.  The sequences it produces are repetitive (the price of peaked
    logits without training); the generic-data parity cases keep the plain weights."""
    w = dict(make_weights(cfg, seed))
    w["model.decoder.embed_tokens.weight"] = np.ascontiguousarray(w["model.decoder.embed_tokens.weight"] * np.float32(gain))
    return w


# ---------------------------------------------------------------------------
# Streaming architectures (reference core/moonshine-streaming-model.{h,cpp}; float definition
# transformers/models/moonshine_streaming/modeling_moonshine_streaming.py)
# ---------------------------------------------------------------------------
@dataclass(frozen=True)
class StreamingArchConfig:
    """Dimensions of a streaming Moonshine model (the keys of ``streaming_config.json``, reference
    ``core/moonshine-streaming-model.cpp:97-113``, plus what the HF config adds).

    ``tiny_streaming`` = HF defaults (``configuration_moonshine_streaming.py:51-69,106-137``).
    The reference does not hold the medium model's dimensions (they come from its
    ``streaming_config.json`` at load; the only hint is the 32768 x 640 head in
    ``lora/export.py:199``): ``medium_streaming`` below is an ASSUMED medium-like shape used for
    benchmarking only -- the engine reads every dimension from the config / tensor shapes.
    """

    name: str
    enc_dim: int
    enc_ffn: int
    enc_layers: int
    enc_heads: int
    dec_dim: int
    dec_ffn: int
    depth: int               # decoder layers
    heads: int               # decoder heads
    windows: tuple           # per encoder layer (past, future), both inclusive (lora/export.py:119-122)
    vocab: int = 32768
    bos: int = 1
    eos: int = 2
    frame_len: int = 80      # 5 ms @ 16 kHz
    max_pos: int = 4096      # pos_emb rows / max_position_embeddings
    max_seq_len: int = 448   # streaming-model.cpp:111-113 default
    rope_theta: float = 10000.0
    partial_rotary: float = 0.8

    @property
    def head_dim(self) -> int:
        return self.dec_dim // self.heads

    @property
    def enc_head_dim(self) -> int:
        return self.enc_dim // self.enc_heads

    @property
    def total_lookahead(self) -> int:
        return sum(int(w[1]) for w in self.windows)

    def streaming_config_json(self) -> str:
        """The ``streaming_config.json`` payload lora/export.py:455-463 writes."""
        return json.dumps({
            "encoder_dim": self.enc_dim, "decoder_dim": self.dec_dim, "depth": self.depth,
            "nheads": self.heads, "head_dim": self.head_dim, "vocab_size": self.vocab,
            "bos_id": self.bos, "eos_id": self.eos, "frame_len": self.frame_len,
            "total_lookahead": self.total_lookahead, "d_model_frontend": self.enc_dim,
            "c1": self.enc_dim * 2, "c2": self.enc_dim, "max_seq_len": self.max_seq_len,
            # additive keys the MI355X engine needs and the ONNX graphs had baked in
            "encoder_heads": self.enc_heads, "windows": [list(w) for w in self.windows],
            "rope_theta": self.rope_theta, "partial_rotary_factor": self.partial_rotary,
        }, indent=2)


_W6 = ((16, 4), (16, 4), (16, 0), (16, 0), (16, 4), (16, 4))
STREAMING_ARCHS = {
    "tiny_streaming": StreamingArchConfig("tiny_streaming", 320, 1280, 6, 8, 320, 1280, 6, 8, _W6),
    "medium_streaming": StreamingArchConfig(
        "medium_streaming", 768, 3072, 12, 12, 640, 2560, 12, 8,
        ((16, 4), (16, 4)) + ((16, 0),) * 8 + ((16, 4), (16, 4))),
    # test-only: encoder_dim != decoder_dim (exercises decoder.proj) and an odd rotary dim
    "micro_streaming": StreamingArchConfig(
        "micro_streaming", 64, 128, 2, 4, 96, 192, 2, 4, ((16, 4), (16, 0)), vocab=512, max_pos=512),
}


def streaming_tensor_specs(cfg: StreamingArchConfig):
    """(name, shape, kind, fan_in) in HF state_dict order (modeling_moonshine_streaming.py:283-296,
    185-207, 133-139, 595-604, 455-456, 782-794, 1005-1010)."""
    De, Fe, Dd, Fd, V = cfg.enc_dim, cfg.enc_ffn, cfg.dec_dim, cfg.dec_ffn, cfg.vocab
    specs = [
        ("model.encoder.embedder.comp.log_k", (), "log_k", 0),
        ("model.encoder.embedder.conv1.weight", (2 * De, De, 5), "matrix", 5 * De),
        ("model.encoder.embedder.conv1.bias", (2 * De,), "bias", 0),
        ("model.encoder.embedder.conv2.weight", (De, 2 * De, 5), "matrix", 10 * De),
        ("model.encoder.embedder.conv2.bias", (De,), "bias", 0),
        ("model.encoder.embedder.linear.weight", (De, cfg.frame_len), "matrix", cfg.frame_len),
    ]
    for l in range(cfg.enc_layers):
        p = f"model.encoder.layers.{l}."
        for n in ("q", "k", "v", "o"):
            specs.append((p + f"self_attn.{n}_proj.weight", (De, De), "matrix", De))
        specs += [
            (p + "mlp.fc1.weight", (Fe, De), "matrix", De),
            (p + "mlp.fc1.bias", (Fe,), "bias", 0),
            (p + "mlp.fc2.weight", (De, Fe), "matrix", Fe),
            (p + "mlp.fc2.bias", (De,), "bias", 0),
            (p + "input_layernorm.gamma", (De,), "gamma0", 0),
            (p + "post_attention_layernorm.gamma", (De,), "gamma0", 0),
        ]
    specs.append(("model.encoder.final_norm.gamma", (De,), "gamma0", 0))
    specs.append(("model.decoder.embed_tokens.weight", (V, Dd), "matrix", Dd))
    for l in range(cfg.depth):
        p = f"model.decoder.layers.{l}."
        for a in ("self_attn", "encoder_attn"):
            for n in ("q", "k", "v", "o"):
                specs.append((p + f"{a}.{n}_proj.weight", (Dd, Dd), "matrix", Dd))
        specs += [
            (p + "mlp.fc1.weight", (2 * Fd, Dd), "matrix", Dd),
            (p + "mlp.fc1.bias", (2 * Fd,), "bias", 0),
            (p + "mlp.fc2.weight", (Dd, Fd), "matrix", Fd),
            (p + "mlp.fc2.bias", (Dd,), "bias", 0),
            (p + "input_layernorm.weight", (Dd,), "scale", 0),
            (p + "post_attention_layernorm.weight", (Dd,), "scale", 0),
            (p + "final_layernorm.weight", (Dd,), "scale", 0),
        ]
    specs.append(("model.decoder.norm.weight", (Dd,), "scale", 0))
    specs.append(("model.decoder.pos_emb.weight", (cfg.max_pos, De), "posemb", 0))
    if De != Dd:
        specs.append(("model.decoder.proj.weight", (Dd, De), "matrix", De))
    specs.append(("proj_out.weight", (V, Dd), "matrix", Dd))
    return specs


def make_streaming_weights(cfg: StreamingArchConfig, seed: int = 0) -> dict[str, np.ndarray]:
    """Deterministic synthetic fp32 weights ("fanin-v1" as above).  Extra kinds: ``gamma0`` = the
    unit-offset LayerNorm scale, stored as gamma - 1 (modeling_moonshine_streaming.py:120-130) ~ 0.1 N;
    ``posemb`` ~ 0.5 N; ``log_k`` = log 0.75 (the module's init).  ``proj_out`` is a separate
    (untied) matrix: tie_word_embeddings defaults to False for this family."""
    out: dict[str, np.ndarray] = {}
    for idx, (name, shape, kind, fan_in) in enumerate(streaming_tensor_specs(cfg)):
        rng = np.random.Generator(np.random.PCG64([seed, 7000 + idx]))
        if kind == "log_k":
            out[name] = np.asarray(np.log(0.75), dtype=np.float32)
            continue
        x = rng.standard_normal(shape, dtype=np.float32)
        if kind == "matrix":
            x *= np.float32(1.0 / np.sqrt(fan_in))
        elif kind == "bias":
            x *= np.float32(0.1)
        elif kind == "scale":
            x = np.float32(1.0) + np.float32(0.1) * x
        elif kind == "gamma0":
            x *= np.float32(0.1)
        elif kind == "posemb":
            x *= np.float32(0.5)
        out[name] = np.ascontiguousarray(x, dtype=np.float32)
    return out


def write_streaming_model_dir(path: str, cfg: StreamingArchConfig, seed: int = 0,
                              weights: dict[str, np.ndarray] | None = None) -> dict[str, np.ndarray]:
    """``model.safetensors`` + ``streaming_config.json`` + ``tokenizer.bin``: the streaming model
    directory of the MI355X engine (the reference's holds five ``.ort`` graphs instead of the
    safetensors file, core/moonshine-streaming-model.cpp:233-300)."""
    import os

    os.makedirs(path, exist_ok=True)
    w = weights if weights is not None else make_streaming_weights(cfg, seed)
    save_safetensors(os.path.join(path, "model.safetensors"), w,
                     {"arch": cfg.name, "format": "moonshine-streaming-hf-f32", "seed": str(seed)})
    with open(os.path.join(path, "streaming_config.json"), "w") as f:
        f.write(cfg.streaming_config_json())
    write_synthetic_tokenizer(os.path.join(path, "tokenizer.bin"), cfg.vocab)
    return w


def make_audio(index: int, n_samples: int = 160000) -> np.ndarray:
    """Synthetic clip ``index``: white noise sigma 0.1 clipped to [-1, 1]
    (BASELINE.md section 3 / SURVEY.md section 8d)."""
    x = np.random.default_rng(1234 + index).standard_normal(n_samples).astype(np.float32) * np.float32(0.1)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


# ---------------------------------------------------------------------------
# safetensors (https://github.com/huggingface/safetensors format v0.x):
#   u64 little-endian header length | JSON header | raw little-endian tensor bytes
# ---------------------------------------------------------------------------
_DT = {"F32": np.float32, "F16": np.float16, "I64": np.int64, "I32": np.int32}


def save_safetensors(path: str, tensors: dict[str, np.ndarray], metadata: dict[str, str] | None = None) -> None:
    header: dict = {}
    if metadata:
        header["__metadata__"] = {str(k): str(v) for k, v in metadata.items()}
    off = 0
    for name, arr in tensors.items():
        assert arr.dtype == np.float32, name
        nbytes = arr.size * 4
        header[name] = {"dtype": "F32", "shape": list(arr.shape), "data_offsets": [off, off + nbytes]}
        off += nbytes
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for arr in tensors.values():
            f.write(np.ascontiguousarray(arr).tobytes())


def load_safetensors(path: str) -> tuple[dict[str, np.ndarray], dict[str, str]]:
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n))
        blob = f.read()
    meta = header.pop("__metadata__", {})
    out = {}
    for name, info in header.items():
        s, e = info["data_offsets"]
        out[name] = np.frombuffer(blob[s:e], dtype=_DT[info["dtype"]]).reshape(info["shape"]).copy()
    return out, meta


def write_model_dir(path: str, cfg: ArchConfig, seed: int = 0, weights: dict[str, np.ndarray] | None = None) -> dict[str, np.ndarray]:
    """Write ``model.safetensors`` + ``tokenizer.bin`` -- the model directory
    contract of the MI355X engine."""
    import os

    os.makedirs(path, exist_ok=True)
    w = weights if weights is not None else make_weights(cfg, seed)
    save_safetensors(
        os.path.join(path, "model.safetensors"),
        w,
        {"arch": cfg.name, "format": "moonshine-hf-f32", "seed": str(seed)},
    )
    write_synthetic_tokenizer(os.path.join(path, "tokenizer.bin"), cfg.vocab)
    return w


# ---------------------------------------------------------------------------
# tokenizer.bin: length-prefixed byte strings (reference core/bin-tokenizer/bin-tokenizer.cpp:46-66,
# scripts/convert_tokenizer.py:29-48): length < 128 -> one byte; otherwise
# first = (len % 128) + 128, second = len // 128; a zero byte is an empty entry.
# ---------------------------------------------------------------------------
SPACE = "\u2581".encode("utf-8")  # sentencepiece word-boundary marker


def encode_tokenizer_bin(tokens: list[bytes]) -> bytes:
    out = bytearray()
    for t in tokens:
        n = len(t)
        if n == 0:
            out.append(0)
        elif n < 128:
            out.append(n)
        else:
            assert n < 128 * 256
            out.append((n % 128) + 128)
            out.append(n // 128)
        out += t
    return bytes(out)


def synthetic_vocab(vocab: int) -> list[bytes]:
    """ids 0,1,2 = <unk>,<s>,</s>; then 256 byte fallbacks ``<0xNN>``; then word pieces (layout of
    the shipped vocabularies, reference core/bin-tokenizer/bin-tokenizer-test.cpp:12-25).  Pieces are
    deterministic: every 3rd one starts a new word; a few are long (>127 bytes) or carry multi-byte /
    deliberately broken UTF-8 to exercise the 2-byte length prefix and sanitize_text."""
    toks = [b"<unk>", b"<s>", b"</s>"]
    toks += [("<0x%02X>" % b).encode() for b in range(256)]
    letters = b"abcdefghijklmnopqrstuvwxyz"
    while len(toks) < vocab:
        k = len(toks)
        n = 1 + (k * 7) % 5
        body = bytes(letters[(k * 31 + j * 17) % 26] for j in range(n))
        if k % 3 == 0:
            body = SPACE + body
        if k == 259:
            body = SPACE                          # the bare word-start marker, so " word" is always spellable
        if k % 997 == 0:
            body = body * 40                      # > 127 bytes: 2-byte length prefix
        elif k % 1013 == 0:
            body = "\u00e9\u4e2d".encode("utf-8") + body  # valid multi-byte
        elif k % 1021 == 0:
            body = b"\xe4\xb8" + body             # truncated 3-byte sequence
        elif k % 1031 == 0:
            body = b"\x9f" + body                 # stray continuation byte
        toks.append(body)
    return toks[:vocab]


def write_synthetic_tokenizer(path: str, vocab: int) -> list[bytes]:
    toks = synthetic_vocab(vocab)
    with open(path, "wb") as f:
        f.write(encode_tokenizer_bin(toks))
    return toks


# ---------------------------------------------------------------------------
# Silero VAD (the `silero_vad.safetensors` the C++ host layer reads, csrc/silero_vad.cpp): synthetic weights with the
# published model's parameter names, for benchmarks and tests that need `vad_threshold > 0` without the real file.
# ---------------------------------------------------------------------------
SILERO_STATE = 128
SILERO_CONV = [(129, 128, 1), (128, 64, 2), (64, 64, 2), (64, 128, 1)]  # (in, out, stride), kernel 3, padding 1


def silero_stft_basis() -> np.ndarray:
    """[258, 1, 256]: hann-windowed DFT rows, real then imaginary (the published model's forward_basis_buffer)."""
    n = 256
    k = np.arange(129)[:, None]
    t = np.arange(n)[None, :]
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)  # periodic hann
    re = np.cos(2 * np.pi * k * t / n) * win
    im = -np.sin(2 * np.pi * k * t / n) * win
    return np.concatenate([re, im])[:, None, :].astype(np.float32)


def make_silero_weights(seed: int = 0) -> dict[str, np.ndarray]:
    """Fan-in scaled random weights so that gates / logits are O(1)."""
    f32 = np.float32
    rng = np.random.default_rng(seed)
    w = {"stft.forward_basis_buffer": silero_stft_basis()}
    for i, (cin, cout, _) in enumerate(SILERO_CONV):
        w[f"encoder.{i}.reparam_conv.weight"] = (rng.standard_normal((cout, cin, 3)) * (1.6 / np.sqrt(cin * 3))).astype(f32)
        w[f"encoder.{i}.reparam_conv.bias"] = (rng.standard_normal(cout) * 0.1).astype(f32)
    for n in ("weight_ih", "weight_hh"):
        w[f"decoder.rnn.{n}"] = (rng.standard_normal((4 * SILERO_STATE, SILERO_STATE)) * (1.5 / np.sqrt(SILERO_STATE))).astype(f32)
    for n in ("bias_ih", "bias_hh"):
        w[f"decoder.rnn.{n}"] = (rng.standard_normal(4 * SILERO_STATE) * 0.1).astype(f32)
    w["decoder.decoder.2.weight"] = (rng.standard_normal((1, SILERO_STATE, 1)) * 0.8).astype(f32)
    w["decoder.decoder.2.bias"] = np.asarray([0.1], f32)
    return w

