"""TEST / BENCH INFRASTRUCTURE ONLY (never on the product path): the CPU baseline SURVEY.md section 8d names as the runnable
stand-in for the reference's CPU-ORT path -- HuggingFace ``MoonshineForConditionalGeneration`` in fp32 eager mode on the
host cores, holding the same synthetic weights, fed the same clips, greedy for the same forced number of steps.

The reference's own transcriber (ONNX Runtime 1.23.2 + int8 .ort graphs) cannot run here: neither the runtime library nor
the graphs are in the checkout.  HF Moonshine is the float definition the reference names as its
oracle (docs/models/accuracy.md:14-19), so this is "the same arithmetic in fp32 on a tuned CPU BLAS" -- slower per FLOP
than int8 ORT kernels, which is stated next to the number wherever it is quoted.

``run(cfg, weights, clips, steps, batch, threads, int8=False)`` -> (token lists, seconds); int8 = the Linear layers with int8
weights and dynamic int8 activations (quantize_linears_int8), the arithmetic class of the reference's shipped graphs.  Clips of one call must have equal length
(they are batched without an attention mask, like the reference's fixed batch of one).
"""
from __future__ import annotations

import time

import numpy as np


def build_hf(cfg, w):
    import torch
    from transformers import MoonshineConfig, MoonshineForConditionalGeneration

    hcfg = MoonshineConfig(
        vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.ffn,
        encoder_num_hidden_layers=cfg.enc_layers, decoder_num_hidden_layers=cfg.dec_layers,
        encoder_num_attention_heads=cfg.heads, decoder_num_attention_heads=cfg.heads)
    hcfg._attn_implementation = "eager"
    m = MoonshineForConditionalGeneration(hcfg).eval()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in k or "inv_freq" in k for k in missing), missing
    return m


def greedy(model, cfg, clips: np.ndarray, steps: int) -> list[list[int]]:
    """clips [B, n] fp32 -> B token lists (BOS + `steps` ids, EOS ignored): encoder once, then one decoder call per
    step on the KV cache -- the loop of reference core/moonshine-model.cpp:380-517 with a batch dimension."""
    import torch
    from transformers.cache_utils import DynamicCache, EncoderDecoderCache

    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(clips))
        enc_out = model.model.encoder(x)
        past = EncoderDecoderCache(DynamicCache(config=model.config), DynamicCache(config=model.config))
        B = x.shape[0]
        cur = torch.full((B, 1), cfg.bos, dtype=torch.long)
        toks = [cur]
        for _ in range(steps):
            out = model(decoder_input_ids=cur, encoder_outputs=enc_out, past_key_values=past, use_cache=True)
            past = out.past_key_values
            cur = out.logits[:, -1].argmax(dim=-1, keepdim=True)   # lowest index among ties, like the reference scan
            toks.append(cur)
        return torch.cat(toks, dim=1).tolist()


def quantize_linears_int8(model):
    """Every nn.Linear with int8 weights and dynamically quantised int8 activations (torch's x86 / fbgemm kernels): the closest
    thing this image can run to what the reference ships -- ONNX Runtime graphs with int8 weights and dynamic int8 activations
    in their MatMuls (core/ort-utils/ort-utils.cpp:256-288 loads them) -- convolutions, norms,
    softmax and the attention products stay fp32."""
    import warnings

    import torch

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.ao.quantization.quantize_dynamic(model, {torch.nn.Linear}, dtype=torch.qint8)


def run(cfg, w, clips: list[np.ndarray], steps: int, batch: int, threads: int, int8: bool = False):
    import torch

    torch.set_num_threads(max(1, threads))
    model = build_hf(cfg, w)
    if int8:
        model = quantize_linears_int8(model)
    greedy(model, cfg, np.stack(clips[:1])[:, :32000], 2)   # warm up the thread pool / allocator
    out = []
    t0 = time.perf_counter()
    for i in range(0, len(clips), batch):
        out += greedy(model, cfg, np.stack(clips[i:i + batch]), steps)
    return out, time.perf_counter() - t0


def teacher_forced_logits(model, clips: np.ndarray, teacher: np.ndarray, sub_batch: int = 32) -> np.ndarray:
    """clips [B, n] fp32, teacher [B, S + 1] ids (BOS first) -> logits [B, S, V] fp32: logits[b, i] is what the decoder
    predicts after consuming teacher[b, :i + 1] -- the quantity step i of the reference loop takes its argmax of
    (core/moonshine-model.cpp:380-517), computed for all positions in ONE causal decoder call per sub-batch (no cascade:
    every position sees the teacher's ids, whatever the model itself would have picked)."""
    import torch

    out = []
    with torch.no_grad():
        for i in range(0, clips.shape[0], sub_batch):
            x = torch.from_numpy(np.ascontiguousarray(clips[i:i + sub_batch]))
            ids = torch.from_numpy(np.ascontiguousarray(teacher[i:i + sub_batch, :-1]).astype(np.int64))
            enc_out = model.model.encoder(x)
            res = model(decoder_input_ids=ids, encoder_outputs=enc_out, use_cache=False)
            out.append(res.logits.float().numpy())
    return np.concatenate(out, axis=0)
