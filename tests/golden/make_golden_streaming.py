"""Generate golden vectors for the streaming path from the reference's own float definition.

Run HERE (needs /root/reference and transformers; neither exists on the GPU box):

    python tests/golden/make_golden_streaming.py

It imports the five graph wrapper modules (Frontend / Encoder / Adapter / CrossKV / DecoderKV) from
the reference's ``language-bindings/python/src/moonshine_voice/lora/export.py`` -- the modules the
reference exports to the ONNX graphs its C++ runtime executes -- wraps a HuggingFace
``MoonshineStreamingForConditionalGeneration`` holding our synthetic weights, and drives them with
the chunk / window schedule of ``core/moonshine-streaming-model.cpp`` (1280-sample chunks,
window = [emitted - 16*depth, total), lookahead held back until final).  Outputs go to
``tests/golden/golden_stream_<case>.npz``; ``tests/test_oracle_streaming.py`` pins the numpy oracle
against them and the GPU tests compare the engine with both.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moonshine_amd.synth import STREAMING_ARCHS, make_audio, make_streaming_weights  # noqa: E402

EXPORT = "/root/reference/language-bindings/python/src/moonshine_voice/lora/export.py"


def load_export():
    spec = importlib.util.spec_from_file_location("ref_lora_export", EXPORT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_hf(cfg, weights):
    from transformers import MoonshineStreamingConfig, MoonshineStreamingForConditionalGeneration

    hc = MoonshineStreamingConfig(
        vocab_size=cfg.vocab, hidden_size=cfg.dec_dim, intermediate_size=cfg.dec_ffn,
        num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads, num_key_value_heads=cfg.heads,
        max_position_embeddings=cfg.max_pos, bos_token_id=cfg.bos, eos_token_id=cfg.eos,
        encoder_config=dict(hidden_size=cfg.enc_dim, intermediate_size=cfg.enc_ffn,
                            num_hidden_layers=cfg.enc_layers, num_attention_heads=cfg.enc_heads,
                            num_key_value_heads=cfg.enc_heads, sliding_windows=[list(w) for w in cfg.windows],
                            max_position_embeddings=cfg.max_pos))
    assert abs(hc.rope_parameters["partial_rotary_factor"] - cfg.partial_rotary) < 1e-9
    model = MoonshineStreamingForConditionalGeneration(hc).eval()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "inv_freq" in m for m in missing), missing
    return model, hc


class Graphs:
    """The five exported graphs as torch modules + the C++ driver's schedule."""

    def __init__(self, cfg, weights):
        ex = load_export()
        self.cfg = cfg
        model, hc = build_hf(cfg, weights)
        enc, dec = model.model.encoder, model.model.decoder
        self.frontend = ex.Frontend(enc.embedder).eval()
        self.encoder = ex.Encoder(enc, [tuple(w) for w in hc.encoder_config.sliding_windows]).eval()
        self.adapter = ex.Adapter(dec).eval()
        self.cross_kv = ex.CrossKV(dec.layers, hc.num_attention_heads, hc.head_dim).eval()
        self.decoder_kv = ex.DecoderKV(dec, model.proj_out, hc.num_attention_heads, hc.head_dim, dec.rotary_emb).eval()
        shapes = ex.frames_state_shapes(enc.embedder, cfg.enc_dim)
        self.state = [torch.zeros(*shapes["sample_buffer"]), torch.zeros(1, dtype=torch.long),
                      torch.zeros(*shapes["conv1_buffer"]), torch.zeros(*shapes["conv2_buffer"]),
                      torch.zeros(1, dtype=torch.long)]
        self.features = torch.zeros(1, 0, cfg.enc_dim)
        self.emitted = 0
        self.pos_offset = 0
        self.memory = torch.zeros(1, 0, cfg.dec_dim)
        self.last_window = None

    @torch.no_grad()
    def chunk(self, audio):
        out = self.frontend(torch.from_numpy(audio)[None], *self.state)
        self.features = torch.cat([self.features, out[0]], dim=1)
        self.state = list(out[1:])

    @torch.no_grad()
    def encode(self, is_final):
        cfg = self.cfg
        total = self.features.shape[1]
        stable = total if is_final else max(0, total - cfg.total_lookahead)
        new = stable - self.emitted
        if new <= 0:
            return 0
        start = max(0, self.emitted - 16 * cfg.depth)
        encoded = self.encoder(self.features[:, start:])
        self.last_window = (start, encoded[0].numpy().copy())
        s0 = self.emitted - start
        mem = self.adapter(encoded[:, s0:s0 + new].clone(), torch.tensor([self.pos_offset]))
        self.memory = torch.cat([self.memory, mem], dim=1)
        self.emitted = stable
        self.pos_offset += new
        return new

    @torch.no_grad()
    def decode(self, tokens, k_self=None, v_self=None):
        cfg = self.cfg
        kc, vc = self.cross_kv(self.memory)
        if k_self is None:
            k_self = torch.zeros(cfg.depth, 1, cfg.heads, 0, cfg.head_dim)
            v_self = torch.zeros(cfg.depth, 1, cfg.heads, 0, cfg.head_dim)
        logits, k2, v2, _, _ = self.decoder_kv(torch.tensor([tokens]), k_self, v_self, kc, vc)
        return logits[0].numpy(), k2, v2, kc, vc


def top8(logits):
    idx = np.argsort(-logits, axis=-1, kind="stable")[:, :8]
    return idx.astype(np.int32), np.take_along_axis(logits, idx, axis=-1).astype(np.float32)


def run_case(name, arch, seed, seconds, update_chunks, n_greedy, audio_index):
    cfg = STREAMING_ARCHS[arch]
    w = make_streaming_weights(cfg, seed)
    g = Graphs(cfg, w)
    audio = make_audio(audio_index, int(seconds * 16000))
    n_chunks = audio.shape[0] // 1280
    mem_lens, windows = [], {}
    c = 0
    update = 0
    while c < n_chunks:
        for _ in range(min(update_chunks, n_chunks - c)):
            g.chunk(audio[c * 1280:(c + 1) * 1280])
            c += 1
        final = c >= n_chunks
        g.encode(final)
        mem_lens.append(g.memory.shape[1])
        if update in (1, 3) and g.last_window is not None:
            windows[update] = g.last_window
        update += 1
    out = {
        "arch": arch, "seed": seed, "audio_index": audio_index, "n_samples": audio.shape[0],
        "update_chunks": update_chunks, "mem_lens": np.asarray(mem_lens, np.int32),
        "features": g.features[0].numpy().astype(np.float32),
        "memory": g.memory[0].numpy().astype(np.float32),
    }
    for u, (start, enc) in windows.items():
        out[f"window{u}_start"] = start
        out[f"window{u}_encoded"] = enc.astype(np.float32)
    # greedy tokens from BOS, one token per call (the plain loop of transcriber.cpp:1441-1466)
    toks = [cfg.bos]
    k = v = None
    step_logits = []
    for _ in range(n_greedy):
        lg, k, v, kc, vc = g.decode([toks[-1]], k, v)
        step_logits.append(lg[0])
        toks.append(int(np.argmax(lg[0])))
    out["greedy_tokens"] = np.asarray(toks, np.int32)
    step_logits = np.stack(step_logits)
    # the same tokens through ONE wide call (what decode_full's verify pass does)
    wide, _, _, kc, vc = g.decode(toks[:-1])
    out["wide_vs_step_maxabs"] = float(np.abs(wide - step_logits).max())
    ti, tv = top8(wide)
    out["wide_top8_idx"], out["wide_top8_val"] = ti, tv
    out["wide_logits_sel"] = wide[:, :64].astype(np.float32)
    if cfg.vocab <= 1024:
        out["wide_logits"] = wide.astype(np.float32)
    out["k_cross_l0_h0"] = kc[0, 0, 0].numpy().astype(np.float32)
    out["v_cross_last_h1"] = vc[-1, 0, 1].numpy().astype(np.float32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"golden_stream_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "features", out["features"].shape, "memory", out["memory"].shape, "mem_lens", mem_lens,
          "tokens", toks[:8], "wide-vs-step", out["wide_vs_step_maxabs"], os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    cases = [("micro_2s", "micro_streaming", 11, 2.4, 5, 12, 3),
             ("tiny_3s", "tiny_streaming", 5, 3.2, 8, 10, 4),
             # the dims bench.py's streaming workload (BASELINE config 5) runs at
             ("medium_2s", "medium_streaming", 7, 2.4, 6, 8, 5)]
    for case in cases:
        if not only or case[0] in only:
            run_case(*case)
