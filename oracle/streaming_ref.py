"""Numpy fp32 restatement of the streaming Moonshine hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference runs five ONNX graphs whose float definitions are the wrapper modules in
``language-bindings/python/src/moonshine_voice/lora/export.py`` (cited ``exp:<line>``) over the
HuggingFace ``modeling_moonshine_streaming.py`` modules (cited ``hf:<line>``); the driver around
them is ``core/moonshine-streaming-model.cpp`` (cited ``ref:<line>``) and its caller
``Transcriber::transcribe_segment_with_streaming_model`` (``core/transcriber.cpp:1311-1487``,
cited ``tr:<line>``).  Arithmetic here follows the former, control flow the latter, all in float32.

Pinned against ``tests/golden/golden_stream_*.npz`` (outputs of those very wrapper modules, made
by ``tests/golden/make_golden_streaming.py``) in ``tests/test_oracle_streaming.py``.
"""
from __future__ import annotations

import math

import numpy as np

from .moonshine_ref import F32, argmax_first, gelu, layer_norm_nobias, silu, softmax_f32
from .weights import StreamingArchConfig


# --- primitives ---------------------------------------------------------------------------------
def frame_cmvn(x: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    """hf:70-79: per-frame mean / rms normalisation (population variance, eps inside the sqrt)."""
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    c = x - mean
    rms = np.sqrt((c * c).mean(axis=-1, keepdims=True, dtype=F32) + F32(eps))
    return (c / rms).astype(F32)


def asinh_comp(x: np.ndarray, log_k: np.ndarray) -> np.ndarray:
    """hf:82-88: asinh(exp(log_k) * x)."""
    return np.arcsinh(np.exp(F32(log_k)) * x).astype(F32)


def unit_offset_layer_norm(x: np.ndarray, gamma: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """hf:120-130: LayerNorm without affine, then * (gamma + 1)."""
    return layer_norm_nobias(x, (gamma + F32(1.0)).astype(F32), eps)


def conv_rows(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int) -> np.ndarray:
    """Valid 1-d convolution over rows: x [L, Cin], w [Cout, Cin, K] -> [Lout, Cout]
    (F.conv1d on the transposed tensors, exp:79-83)."""
    L = x.shape[0]
    cout, cin, K = w.shape
    lout = (L - K) // stride + 1
    if lout <= 0:
        raise ValueError("conv input shorter than the kernel (the reference graph fails the same way)")
    idx = np.arange(lout)[:, None] * stride + np.arange(K)[None, :]       # [Lout, K]
    cols = x[idx]                                                         # [Lout, K, Cin]
    wk = np.transpose(w, (0, 2, 1)).reshape(cout, K * cin)                # [(K,Cin)] per output
    return (cols.reshape(lout, K * cin) @ wk.T + b[None, :]).astype(F32)


def rope_tables(cfg: StreamingArchConfig, positions: np.ndarray):
    """hf:486-524 + hf:552-557: inv_freq over dim = int(head_dim * partial_rotary_factor);
    cos/sin of the first half of cat(freqs, freqs) repeated pairwise -> rotary width 2*ceil(dim/2)."""
    dim = int(cfg.head_dim * cfg.partial_rotary)
    inv = (1.0 / (cfg.rope_theta ** (np.arange(0, dim, 2, dtype=np.float32) / F32(dim)))).astype(F32)
    fr = positions.astype(F32)[:, None] * inv[None, :]                   # [n, nfreq]
    emb = np.concatenate([fr, fr], axis=-1)
    half = emb.shape[-1] // 2
    cos = np.repeat(np.cos(emb)[:, :half], 2, axis=-1).astype(F32)       # [n, rotary_dim]
    sin = np.repeat(np.sin(emb)[:, :half], 2, axis=-1).astype(F32)
    return cos, sin


def apply_rope(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """hf:527-571, interleaved pairs on the first rotary_dim dims; x [H, n, dh]."""
    r = cos.shape[-1]
    xr, xp = x[..., :r], x[..., r:]
    x1, x2 = xr[..., 0::2], xr[..., 1::2]
    rot = np.stack([-x2, x1], axis=-1).reshape(xr.shape)
    return np.concatenate([xr * cos[None] + rot * sin[None], xp], axis=-1).astype(F32)


# --- the five graphs ------------------------------------------------------------------------------
class FrontendState:
    """Carry-over state of the frontend graph (ref:36-43, exp:42-50)."""

    def __init__(self, cfg: StreamingArchConfig):
        self.sample_buffer = np.zeros(cfg.frame_len - 1, F32)
        self.sample_len = 0
        self.conv1_buffer = np.zeros((4, cfg.enc_dim), F32)       # last 4 pre-conv1 frames [frame, C]
        self.conv2_buffer = np.zeros((4, 2 * cfg.enc_dim), F32)   # last 4 conv1 outputs
        self.frame_count = 0


def frontend(w, cfg: StreamingArchConfig, st: FrontendState, chunk: np.ndarray) -> np.ndarray:
    """exp:67-97 (Frontend.forward): audio chunk + state -> features [n, enc_dim]; updates ``st``."""
    p = "model.encoder.embedder."
    buffered = st.sample_buffer.shape[0]
    keep = min(int(st.sample_len), buffered)
    combined = np.concatenate([st.sample_buffer[:keep], chunk.astype(F32)])
    nf = combined.shape[0] // cfg.frame_len
    used = nf * cfg.frame_len
    frames = combined[:used].reshape(nf, cfg.frame_len)
    hidden = silu(asinh_comp(frame_cmvn(frames), w[p + "comp.log_k"]) @ w[p + "linear.weight"].T)   # [nf, De]
    conv1_in = np.concatenate([st.conv1_buffer, hidden], axis=0)
    conv1_out = silu(conv_rows(conv1_in, w[p + "conv1.weight"], w[p + "conv1.bias"], 2))
    conv2_in = np.concatenate([st.conv2_buffer, conv1_out], axis=0)
    features = conv_rows(conv2_in, w[p + "conv2.weight"], w[p + "conv2.bias"], 2)
    rem = combined[used:]
    st.sample_buffer = np.concatenate([rem, np.zeros(buffered, F32)])[:buffered]
    st.sample_len = rem.shape[0]
    st.conv1_buffer = hidden[-4:].copy()       # like the graph, assumes >= 4 new frames per call
    st.conv2_buffer = conv1_out[-4:].copy()
    st.frame_count += nf
    return features


def encoder(w, cfg: StreamingArchConfig, feats: np.ndarray) -> np.ndarray:
    """exp:113-127 (Encoder.forward): per-layer inclusive (past, future) window masks over the rows it
    is given; layers hf:251-280 (pre-LN, MHA without RoPE or bias, GELU MLP with bias); final norm."""
    h = feats.astype(F32)
    L = h.shape[0]
    H, dh = cfg.enc_heads, cfg.enc_head_dim
    pos = np.arange(L)
    dist = pos[:, None] - pos[None, :]
    scale = F32(dh ** -0.5)
    neg = np.finfo(np.float32).min
    for l, (past, future) in enumerate(cfg.windows):
        p = f"model.encoder.layers.{l}."
        allowed = (dist >= -future) & (dist <= past)
        x = unit_offset_layer_norm(h, w[p + "input_layernorm.gamma"])
        q = (x @ w[p + "self_attn.q_proj.weight"].T).reshape(L, H, dh).transpose(1, 0, 2)
        k = (x @ w[p + "self_attn.k_proj.weight"].T).reshape(L, H, dh).transpose(1, 0, 2)
        v = (x @ w[p + "self_attn.v_proj.weight"].T).reshape(L, H, dh).transpose(1, 0, 2)
        s = (q @ k.transpose(0, 2, 1)) * scale + np.where(allowed, F32(0), F32(neg))[None]
        a = (softmax_f32(s.astype(F32)) @ v).transpose(1, 0, 2).reshape(L, H * dh)
        h = h + a @ w[p + "self_attn.o_proj.weight"].T
        x = unit_offset_layer_norm(h, w[p + "post_attention_layernorm.gamma"])
        x = gelu(x @ w[p + "mlp.fc1.weight"].T + w[p + "mlp.fc1.bias"])
        h = (h + x @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"]).astype(F32)
    return unit_offset_layer_norm(h, w["model.encoder.final_norm.gamma"])


def adapter(w, cfg: StreamingArchConfig, encoded: np.ndarray, pos_offset: int) -> np.ndarray:
    """exp:141-144 (Adapter.forward): proj(encoded + pos_emb[pos_offset + i]); proj = Identity when
    the two widths agree (hf:791-794)."""
    n = encoded.shape[0]
    x = encoded + w["model.decoder.pos_emb.weight"][pos_offset:pos_offset + n]
    if "model.decoder.proj.weight" in w:
        x = x @ w["model.decoder.proj.weight"].T
    return x.astype(F32)


def cross_kv(w, cfg: StreamingArchConfig, memory: np.ndarray):
    """exp:161-167 (CrossKV.forward): k, v [depth, H, M, dh]."""
    M = memory.shape[0]
    H, dh = cfg.heads, cfg.head_dim
    ks, vs = [], []
    for l in range(cfg.depth):
        p = f"model.decoder.layers.{l}.encoder_attn."
        ks.append((memory @ w[p + "k_proj.weight"].T).reshape(M, H, dh).transpose(1, 0, 2))
        vs.append((memory @ w[p + "v_proj.weight"].T).reshape(M, H, dh).transpose(1, 0, 2))
    return np.stack(ks).astype(F32), np.stack(vs).astype(F32)


def decoder_kv(w, cfg: StreamingArchConfig, tokens, k_self, v_self, k_cross, v_cross, cross_probs: list | None = None):
    """exp:207-256 (DecoderKV.forward): n tokens + caches -> logits [n, V], grown self caches.
    k_self / v_self: [depth, H, S, dh] (S may be 0).  ``cross_probs``: if a list, one [H, n, M] array of
    cross-attention probabilities per layer is appended -- the ``cross_attentions.{l}`` outputs of the reference's
    decoder_kv_with_attention graph (core/moonshine-streaming-model.cpp:946-1066)."""
    tokens = np.asarray(tokens, dtype=np.int64)
    n = tokens.shape[0]
    cached = k_self.shape[2]
    H, dh = cfg.heads, cfg.head_dim
    scale = F32(dh ** -0.5)
    neg = np.finfo(np.float32).min
    h = w["model.decoder.embed_tokens.weight"][tokens].astype(F32)
    cos, sin = rope_tables(cfg, np.arange(cached, cached + n))
    qpos = np.arange(cached, cached + n)
    kpos = np.arange(cached + n)
    causal = qpos[:, None] >= kpos[None, :]
    new_k, new_v = [], []
    for l in range(cfg.depth):
        p = f"model.decoder.layers.{l}."
        x = layer_norm_nobias(h, w[p + "input_layernorm.weight"])
        q = (x @ w[p + "self_attn.q_proj.weight"].T).reshape(n, H, dh).transpose(1, 0, 2)
        k = (x @ w[p + "self_attn.k_proj.weight"].T).reshape(n, H, dh).transpose(1, 0, 2)
        v = (x @ w[p + "self_attn.v_proj.weight"].T).reshape(n, H, dh).transpose(1, 0, 2)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        k = np.concatenate([k_self[l], k], axis=1)
        v = np.concatenate([v_self[l], v], axis=1)
        new_k.append(k)
        new_v.append(v)
        s = (q @ k.transpose(0, 2, 1)) * scale
        s = np.where(causal[None], s, F32(neg)).astype(F32)
        a = (softmax_f32(s) @ v).transpose(1, 0, 2).reshape(n, H * dh)
        h = h + a @ w[p + "self_attn.o_proj.weight"].T
        x = layer_norm_nobias(h, w[p + "post_attention_layernorm.weight"])
        q = (x @ w[p + "encoder_attn.q_proj.weight"].T).reshape(n, H, dh).transpose(1, 0, 2)
        s = ((q @ k_cross[l].transpose(0, 2, 1)) * scale).astype(F32)
        if cross_probs is not None:
            cross_probs.append(softmax_f32(s))
        a = (softmax_f32(s) @ v_cross[l]).transpose(1, 0, 2).reshape(n, H * dh)
        h = h + a @ w[p + "encoder_attn.o_proj.weight"].T
        x = layer_norm_nobias(h, w[p + "final_layernorm.weight"])
        y = x @ w[p + "mlp.fc1.weight"].T + w[p + "mlp.fc1.bias"]
        val, gate = np.split(y, 2, axis=-1)                                 # hf:459-461
        h = (h + (silu(gate) * val) @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"]).astype(F32)
    h = layer_norm_nobias(h, w["model.decoder.norm.weight"])
    logits = (h @ w["proj_out.weight"].T).astype(F32)                     # exp:254-255
    return logits, np.stack(new_k).astype(F32), np.stack(new_v).astype(F32)


# --- driver: MoonshineStreamingState / MoonshineStreamingModel -----------------------------------------
class StreamState:
    """ref:36-71 (MoonshineStreamingState) with reset() = ref:124-156."""

    def __init__(self, cfg: StreamingArchConfig):
        self.cfg = cfg
        self.reset()

    def reset(self):
        cfg = self.cfg
        self.fe = FrontendState(cfg)
        self.features = np.zeros((0, cfg.enc_dim), F32)      # accumulated_features
        self.encoder_frames_emitted = 0
        self.adapter_pos_offset = 0
        self.memory = np.zeros((0, cfg.dec_dim), F32)
        self.decoder_reset()
        self.k_cross = self.v_cross = None
        self.cross_kv_valid = False

    def decoder_reset(self):
        """ref:1399-1406: the cross K/V stay valid."""
        cfg = self.cfg
        self.k_self = np.zeros((cfg.depth, cfg.heads, 0, cfg.head_dim), F32)
        self.v_self = np.zeros((cfg.depth, cfg.heads, 0, cfg.head_dim), F32)

    @property
    def memory_len(self):
        return self.memory.shape[0]

    @property
    def cache_seq_len(self):
        return self.k_self.shape[2]


def process_audio_chunk(w, cfg, st: StreamState, chunk: np.ndarray) -> int:
    """ref:441-602: run the frontend on one chunk, append its features."""
    if chunk.shape[0] == 0:
        return 0
    f = frontend(w, cfg, st.fe, chunk)
    st.features = np.concatenate([st.features, f], axis=0)
    return f.shape[0]


def encode(w, cfg, st: StreamState, is_final: bool) -> int:
    """ref:604-772: encoder over [emitted - 16*depth, total), adapter over the newly stable frames."""
    total = st.features.shape[0]
    if total == 0:
        return 0
    stable = total if is_final else max(0, total - cfg.total_lookahead)      # ref:624-626
    new = stable - st.encoder_frames_emitted
    if new <= 0:
        return 0
    start = max(0, st.encoder_frames_emitted - 16 * cfg.depth)               # ref:638-642
    enc = encoder(w, cfg, st.features[start:])
    s0 = st.encoder_frames_emitted - start
    mem = adapter(w, cfg, enc[s0:s0 + new], st.adapter_pos_offset)
    st.memory = np.concatenate([st.memory, mem], axis=0)
    st.cross_kv_valid = False
    st.encoder_frames_emitted = stable
    st.adapter_pos_offset += new
    return new


def _run_decoder(w, cfg, st: StreamState, tokens) -> np.ndarray:
    """ref:867-1082 (run_decoder_with_cross_kv) + the lazy compute_cross_kv of its callers."""
    if not st.cross_kv_valid:
        st.k_cross, st.v_cross = cross_kv(w, cfg, st.memory)                 # ref:779-860
        st.cross_kv_valid = True
    logits, st.k_self, st.v_self = decoder_kv(w, cfg, tokens, st.k_self, st.v_self, st.k_cross, st.v_cross)
    return logits


def decode_tokens(w, cfg, st: StreamState, tokens) -> np.ndarray:
    """ref:1136-1185 (decode_step is the one-token case, ref:1089-1129)."""
    return _run_decoder(w, cfg, st, tokens)


def cross_attention_for_tokens(w, cfg, st: StreamState, tokens) -> np.ndarray:
    """The attention the word-timestamp path aligns (core/transcriber.cpp:1028-1068): probabilities of every layer, head
    and decoder position for the token sequence fed from an empty self cache, as [depth * H, n, memory_len] -- the layout
    align_words takes.  (The reference accumulates them call by call; fed the same tokens the result is the same.)"""
    st.decoder_reset()
    if not st.cross_kv_valid:
        st.k_cross, st.v_cross = cross_kv(w, cfg, st.memory)
        st.cross_kv_valid = True
    probs: list = []
    _, st.k_self, st.v_self = decoder_kv(w, cfg, tokens, st.k_self, st.v_self, st.k_cross, st.v_cross, probs)
    return np.concatenate(probs, axis=0).astype(F32)   # [L][H, n, M] -> [L*H, n, M]


def max_tokens_for_memory(cfg, memory_len: int) -> int:
    """ref:1217-1219: float32 duration, double product."""
    duration = F32(memory_len) * F32(0.020)
    return min(int(math.ceil(float(duration) * 6.5)), cfg.max_seq_len)


def decode_full(w, cfg, st: StreamState, draft=None, stats: dict | None = None, biaser=None):
    """ref:1192-1397: greedy decode from BOS; with a draft, one wide verify call, keep the agreeing
    prefix, and on divergence reset the self cache and re-run the accepted prefix before continuing.
    ``biaser`` (oracle.biaser_ref.ContextBiaser, optional) adds its bonuses before every token choice,
    the verify pass included; its walk follows the teacher-forced prefix (ref:1234-1246, 1304-1315,
    1354-1361).  Returns the content tokens (no BOS / EOS).  The caller resets the decoder first (tr:1385)."""
    if st.memory_len == 0:
        return []
    out: list[int] = []
    max_tokens = max_tokens_for_memory(cfg, st.memory_len)
    if biaser is not None:
        biaser.reset()

    def biased_argmax(row: np.ndarray) -> int:
        if biaser is not None:
            row = row.copy()
            biaser.apply(row)
        return argmax_first(row)

    def continue_ar(tok: int):
        cur = tok
        while cur != cfg.eos and len(out) < max_tokens:                      # ref:1275-1276
            out.append(cur)
            if biaser is not None:
                biaser.advance(cur)
            cur = biased_argmax(_run_decoder(w, cfg, st, [cur])[0])

    draft = list(draft) if draft is not None else []
    if draft:
        toks = [cfg.bos] + draft
        logits = _run_decoder(w, cfg, st, toks)
        pred = []
        for t in range(len(toks)):                                           # ref:1307-1315
            pred.append(biased_argmax(logits[t]))
            if biaser is not None and t + 1 < len(toks):
                biaser.advance(toks[t + 1])
        d = 0
        for i in range(len(draft)):                                          # ref:1318-1325
            if pred[i] == draft[i]:
                d = i + 1
            else:
                break
        out.extend(draft[:d])
        if stats is not None:
            stats["accepted"] = d
            stats["draft"] = len(draft)
        if d == len(draft):
            continue_ar(pred[len(draft)])
        else:
            st.decoder_reset()                                               # ref:1338-1340
            logits2 = _run_decoder(w, cfg, st, [cfg.bos] + draft[:d])
            if biaser is not None:                                           # ref:1354-1361
                biaser.reset()
                for i in range(d):
                    biaser.advance(draft[i])
            continue_ar(biased_argmax(logits2[d]))
    else:
        logits = _run_decoder(w, cfg, st, [cfg.bos])
        continue_ar(biased_argmax(logits[0]))
    return out


# --- caller: Transcriber::transcribe_segment_with_streaming_model ---------------------------------------
class SegmentStreamer:
    """tr:1311-1487 for one segment id: feeds only the new audio in 1280-sample chunks (a trailing
    partial chunk waits), encodes, and decodes either with the previous tokens as a draft
    (use_speculative_decoding) or with the plain per-token loop."""

    CHUNK = 1280

    def __init__(self, w, cfg, use_speculative_decoding=True, max_tokens_per_second=6.5):
        self.w, self.cfg = w, cfg
        self.st = StreamState(cfg)
        self.samples_processed = 0
        self.last_tokens: list[int] = []
        self.spec = use_speculative_decoding
        self.mtps = max_tokens_per_second
        self.first = True

    def update(self, audio: np.ndarray, is_final: bool, stats: dict | None = None) -> list[int]:
        """``audio`` is the whole segment so far.  Returns the token list incl. BOS (and EOS if the
        plain loop hit it), i.e. what the reference hands to tokens_to_text."""
        w, cfg, st = self.w, self.cfg, self.st
        is_new = self.first
        self.first = False
        n = audio.shape[0]
        if self.samples_processed < n:
            new = audio[self.samples_processed:]
            cc = new.shape[0] // self.CHUNK
            for c in range(cc):
                process_audio_chunk(w, cfg, st, new[c * self.CHUNK:(c + 1) * self.CHUNK])
            encode(w, cfg, st, is_final)
            self.samples_processed += cc * self.CHUNK
        if st.memory_len == 0:
            return []
        st.decoder_reset()
        duration = F32(n) / F32(16000.0)
        max_tokens = min(int(math.ceil(float(duration * F32(self.mtps)))), 256)   # tr:1388-1392
        if self.spec and not is_new and self.last_tokens:
            draft = [t for t in self.last_tokens if t not in (cfg.bos, cfg.eos)]
            tokens = [cfg.bos] + decode_full(w, cfg, st, draft, stats)
        else:
            tokens = [cfg.bos]
            cur = cfg.bos
            for _ in range(max_tokens):                                       # tr:1441-1466
                nxt = argmax_first(decode_tokens(w, cfg, st, [cur])[0])
                tokens.append(nxt)
                cur = nxt
                if nxt == cfg.eos:
                    break
        self.last_tokens = list(tokens)
        return tokens
