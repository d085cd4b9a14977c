"""Numpy fp32 restatement of the non-streaming Moonshine hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Arithmetic follows the float definition the reference names as its oracle,
HuggingFace ``transformers/models/moonshine/modeling_moonshine.py`` (cited as
``hf:<line>`` below); control flow follows the reference driver
``core/moonshine-model.cpp`` (cited as ``ref:<line>``).  All arithmetic is
float32 with fp32 softmax, like the HF eager path.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf

from .host_ref import max_decode_len
from .weights import ArchConfig

F32 = np.float32


# --- primitives --------------------------------------------------------------
def layer_norm_nobias(x: np.ndarray, w: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """nn.LayerNorm(D, bias=False)  hf:379-380, 538 (eps default 1e-5)."""
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps))).astype(F32) * w


def gelu(x: np.ndarray) -> np.ndarray:
    """exact (erf) GELU: encoder_hidden_act = "gelu"  configuration_moonshine.py:90."""
    return (F32(0.5) * x * (F32(1.0) + erf(x * F32(1.0 / math.sqrt(2.0))))).astype(F32)


def silu(x: np.ndarray) -> np.ndarray:
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def softmax_f32(s: np.ndarray) -> np.ndarray:
    """hf:187  softmax(dim=-1, dtype=float32)."""
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m, dtype=F32)
    return e / e.sum(axis=-1, keepdims=True, dtype=F32)


def conv1d(x: np.ndarray, w: np.ndarray, b: np.ndarray | None, stride: int) -> np.ndarray:
    """x [Cin, L], w [Cout, Cin, K] -> [Cout, Lout]; valid padding (nn.Conv1d)."""
    cin, L = x.shape
    cout, _, K = w.shape
    lout = (L - K) // stride + 1
    # im2col: cols[(ci,k), t] = x[ci, t*stride + k]
    idx = np.arange(lout)[None, :] * stride + np.arange(K)[:, None]          # [K, Lout]
    cols = x[:, idx].reshape(cin * K, lout)                                   # [(Cin,K), Lout]
    y = w.reshape(cout, cin * K) @ cols
    if b is not None:
        y = y + b[:, None]
    return y.astype(F32)


def conv_out_lengths(n_samples: int) -> tuple[int, int, int]:
    """hf:500-508."""
    l1 = int((n_samples - 127) / 64 + 1)
    l2 = int((l1 - 7) / 3 + 1)
    l3 = int((l2 - 3) / 2 + 1)
    return l1, l2, l3


def rope_tables(cfg: ArchConfig, positions: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """cos/sin [P, rotary_dim/2] in fp32.  hf:132-139 (inv_freq), hf:150-154
    (freqs = inv_freq * pos in fp32), hf:225-226 (first half, interleaved-repeated)."""
    dim = cfg.rotary_dim
    inv_freq = (F32(1.0) / (F32(cfg.rope_theta) ** (np.arange(0, dim, 2, dtype=F32) / F32(dim)))).astype(F32)
    freqs = positions.astype(F32)[:, None] * inv_freq[None, :]
    return np.cos(freqs).astype(F32), np.sin(freqs).astype(F32)


def apply_rope(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """x [H, P, dh]; rotate the first 2*len(cos[0]) dims as interleaved pairs
    (x[2j], x[2j+1]) -> (x0*c - x1*s, x1*c + x0*s); remaining dims pass through.
    hf:196-240 (rotate_half uses x[0::2], x[1::2])."""
    nrot = cos.shape[-1] * 2
    out = x.copy()
    x0 = x[..., 0:nrot:2]
    x1 = x[..., 1:nrot:2]
    out[..., 0:nrot:2] = x0 * cos - x1 * sin
    out[..., 1:nrot:2] = x1 * cos + x0 * sin
    return out.astype(F32)


def _heads(x: np.ndarray, H: int) -> np.ndarray:
    """[P, D] -> [H, P, dh]."""
    P, D = x.shape
    return x.reshape(P, H, D // H).transpose(1, 0, 2)


def attention(q: np.ndarray, k: np.ndarray, v: np.ndarray, causal_offset: int | None = None) -> np.ndarray:
    """q [H,Pq,dh], k/v [H,Pk,dh] -> [Pq, H*dh].  hf:171-193, scale dh**-0.5.
    causal_offset: query i may see keys <= causal_offset + i."""
    H, Pq, dh = q.shape
    s = np.matmul(q, k.transpose(0, 2, 1)).astype(F32) * F32(dh ** -0.5)   # batched BLAS (einsum has no BLAS path here)
    if causal_offset is not None:
        Pk = k.shape[1]
        mask = np.arange(Pk)[None, :] > (causal_offset + np.arange(Pq))[:, None]
        s = np.where(mask[None], F32(-np.inf), s)
    p = softmax_f32(s)
    o = np.matmul(p, v).astype(F32)
    return o.transpose(1, 0, 2).reshape(Pq, H * dh)


# --- encoder -----------------------------------------------------------------
def encoder_forward(w: dict, cfg: ArchConfig, audio: np.ndarray, taps: dict | None = None) -> np.ndarray:
    """MoonshineEncoder.forward hf:551-612 for one clip: audio [n] -> [T, D]."""
    pre = "model.encoder."
    x = audio.astype(F32)[None, :]
    h = np.tanh(conv1d(x, w[pre + "conv1.weight"], None, 64)).astype(F32)          # hf:573
    if taps is not None:
        taps["conv1_tanh"] = h.T.copy()
    # GroupNorm(1 group) over (C, L), eps 1e-5, affine per channel  hf:533, 574
    mu = h.mean(dtype=np.float64)
    var = ((h.astype(np.float64) - mu) ** 2).mean()
    h = ((h - F32(mu)) / F32(math.sqrt(var + 1e-5))).astype(F32)
    h = h * w[pre + "groupnorm.weight"][:, None] + w[pre + "groupnorm.bias"][:, None]
    if taps is not None:
        taps["groupnorm"] = h.T.copy()
    h = gelu(conv1d(h, w[pre + "conv2.weight"], w[pre + "conv2.bias"], 3))        # hf:575
    if taps is not None:
        taps["conv2_gelu"] = h.T.copy()
    h = gelu(conv1d(h, w[pre + "conv3.weight"], w[pre + "conv3.bias"], 2))        # hf:576
    h = np.ascontiguousarray(h.T)                                                  # [T, D] hf:577
    if taps is not None:
        taps["conv3_gelu"] = h.copy()
    T = h.shape[0]
    cos, sin = rope_tables(cfg, np.arange(T))                                      # hf:594-595
    H = cfg.heads
    for l in range(cfg.enc_layers):
        p = f"{pre}layers.{l}."
        r = h
        y = layer_norm_nobias(h, w[p + "input_layernorm.weight"])
        q = apply_rope(_heads(y @ w[p + "self_attn.q_proj.weight"].T, H), cos, sin)
        k = apply_rope(_heads(y @ w[p + "self_attn.k_proj.weight"].T, H), cos, sin)
        v = _heads(y @ w[p + "self_attn.v_proj.weight"].T, H)
        a = attention(q, k, v)
        h = r + a @ w[p + "self_attn.o_proj.weight"].T
        r = h
        y = layer_norm_nobias(h, w[p + "post_attention_layernorm.weight"])
        y = gelu(y @ w[p + "mlp.fc1.weight"].T + w[p + "mlp.fc1.bias"])
        h = (r + y @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"]).astype(F32)
        if taps is not None:
            taps[f"enc_layer{l}"] = h.copy()
    return layer_norm_nobias(h, w[pre + "layer_norm.weight"])


# --- decoder -----------------------------------------------------------------
class DecoderState:
    """Self-attention KV cache + frozen cross-attention KV (the ``past_key_values.*``
    tensors of ref:354-368 / the cache rule ref:462-477)."""

    def __init__(self, w: dict, cfg: ArchConfig, enc: np.ndarray):
        self.cfg = cfg
        H = cfg.heads
        self.cross = []
        for l in range(cfg.dec_layers):
            p = f"model.decoder.layers.{l}.encoder_attn."
            # cross K/V are computed once from the encoder output, no RoPE (hf:327)
            self.cross.append((_heads(enc @ w[p + "k_proj.weight"].T, H), _heads(enc @ w[p + "v_proj.weight"].T, H)))
        self.self_k = [np.zeros((H, 0, cfg.head_dim), F32) for _ in range(cfg.dec_layers)]
        self.self_v = [np.zeros((H, 0, cfg.head_dim), F32) for _ in range(cfg.dec_layers)]

    @property
    def past_len(self) -> int:
        return self.self_k[0].shape[1]


def decoder_forward(w: dict, cfg: ArchConfig, st: DecoderState, tokens, cross_probs: list | None = None) -> np.ndarray:
    """MoonshineDecoder.forward hf:639-711 + tied LM head hf:858-: feed ``tokens``
    (n >= 1 ids) at positions past_len .. past_len+n-1, append to the self cache,
    return logits [n, V].  ``cross_probs``: if a list, one [H, n, T] array of cross-attention
    probabilities per layer is appended (the ``cross_attentions.{l}`` outputs of the reference's
    attention-exporting decoder, core/moonshine-model.cpp:480-500)."""
    H = cfg.heads
    ids = np.asarray(tokens, dtype=np.int64).reshape(-1)
    n = ids.shape[0]
    E = w["model.decoder.embed_tokens.weight"]
    h = E[ids].astype(F32)                                                         # hf:665 (no scaling)
    past = st.past_len
    cos, sin = rope_tables(cfg, np.arange(past, past + n))                         # hf:670-673
    for l in range(cfg.dec_layers):
        p = f"model.decoder.layers.{l}."
        r = h
        y = layer_norm_nobias(h, w[p + "input_layernorm.weight"])
        q = apply_rope(_heads(y @ w[p + "self_attn.q_proj.weight"].T, H), cos, sin)
        k = apply_rope(_heads(y @ w[p + "self_attn.k_proj.weight"].T, H), cos, sin)
        v = _heads(y @ w[p + "self_attn.v_proj.weight"].T, H)
        st.self_k[l] = np.concatenate([st.self_k[l], k], axis=1)
        st.self_v[l] = np.concatenate([st.self_v[l], v], axis=1)
        a = attention(q, st.self_k[l], st.self_v[l], causal_offset=past)
        h = r + a @ w[p + "self_attn.o_proj.weight"].T
        r = h
        y = layer_norm_nobias(h, w[p + "post_attention_layernorm.weight"])
        q = _heads(y @ w[p + "encoder_attn.q_proj.weight"].T, H)
        ck, cv = st.cross[l]
        a = attention(q, ck, cv)
        if cross_probs is not None:
            cross_probs.append(softmax_f32(np.einsum("hqd,hkd->hqk", q, ck).astype(F32) * F32(cfg.head_dim ** -0.5)))
        h = r + a @ w[p + "encoder_attn.o_proj.weight"].T
        r = h
        y = layer_norm_nobias(h, w[p + "final_layernorm.weight"])
        y = y @ w[p + "mlp.fc1.weight"].T + w[p + "mlp.fc1.bias"]
        val, gate = np.split(y, 2, axis=-1)                                        # hf:92-96: (value, gate)
        y = silu(gate) * val
        h = (r + y @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"]).astype(F32)
    h = layer_norm_nobias(h, w["model.decoder.norm.weight"])
    return (h @ E.T).astype(F32)                                                   # tied head, hf:836-850


def argmax_first(x: np.ndarray) -> int:
    """First-max-wins linear scan, ref core/ort-utils/moonshine-tensor-view.cpp:222-236.
    np.argmax already returns the lowest index among ties."""
    return int(np.argmax(x))


def greedy_decode(
    w: dict,
    cfg: ArchConfig,
    enc: np.ndarray,
    max_len: int,
    ignore_eos: bool = False,
    return_logits: bool = False,
    teacher: list[int] | None = None,
    return_cross_attention: bool = False,
):
    """The decode loop of MoonshineModel::transcribe ref:370-517: start from BOS,
    one token per step, first-max argmax, stop after appending EOS or after
    ``max_len`` steps.  Returns tokens including BOS (and EOS if emitted).
    ``teacher``: feed these ids instead of the model's own choices (parity tests)."""
    st = DecoderState(w, cfg, enc)
    tokens = [cfg.bos]
    logits_all = []
    cur = cfg.bos
    cross = []  # per step: [L][H, 1, T]
    for i in range(max_len):
        probs = [] if return_cross_attention else None
        logits = decoder_forward(w, cfg, st, [cur], probs)[0]
        if return_cross_attention:
            cross.append(np.stack([p[:, 0, :] for p in probs]))  # [L, H, T]
        if return_logits:
            logits_all.append(logits)
        nxt = argmax_first(logits)
        tokens.append(nxt)
        if nxt == cfg.eos and not ignore_eos:
            break
        cur = nxt if teacher is None else teacher[i + 1] if i + 1 < len(teacher) else nxt
    if return_cross_attention:
        # [L*H, steps, T]: the layout align_words takes (core/moonshine-model.cpp:616-640)
        att = np.stack(cross, axis=2).reshape(cfg.dec_layers * cfg.heads, len(cross), -1).astype(F32)
        return (tokens, np.stack(logits_all), att) if return_logits else (tokens, att)
    if return_logits:
        return tokens, np.stack(logits_all) if logits_all else np.zeros((0, cfg.vocab), F32)
    return tokens


def transcribe_tokens(w: dict, cfg: ArchConfig, audio: np.ndarray, max_tokens_per_second: float = 6.5, **kw):
    """MoonshineModel::transcribe ref:216-563 up to (not including) detokenisation."""
    enc = encoder_forward(w, cfg, audio)
    return greedy_decode(w, cfg, enc, max_decode_len(audio.shape[0], max_tokens_per_second), **kw)
