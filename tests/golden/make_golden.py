"""Generate golden vectors for the oracle from the HuggingFace float implementation.

Run ONCE in the build container (needs ``transformers``; imports nothing from the
repo's product path):

    python tests/golden/make_golden.py

The reference names HF ``transformers`` Moonshine as its float oracle
(reference docs/models/accuracy.md:14-19, scripts/eval-librispeech.py:434-477).
For each case we build the HF model with the architecture's dimensions, load the
deterministic synthetic weights from ``oracle.weights.make_weights`` into it
(so no weight file has to be stored), run encoder + a manual greedy decode loop
with KV cache, and store small slices of the results in ``golden_<case>.npz``:

  enc_rows      indices of the stored encoder frames
  enc           last_hidden_state[enc_rows, :]          fp32
  enc_absmean   mean |last_hidden_state| over the whole tensor
  tokens        greedy ids incl. BOS (EOS ignored, fixed step count)
  logit_idx     per step: indices of the top-8 logits
  logit_val     per step: their values
  conv_rows / conv3   rows of the conv-stem output (pre-transformer)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.weights import ARCHS, make_audio, make_weights  # noqa: E402

CASES = [
    # (case name, arch, weight seed, clip index, n_samples, steps)
    ("micro_1s", "micro", 0, 0, 16000, 7),
    ("micro_ragged", "micro", 1, 3, 23789, 10),
    ("tiny_2s", "tiny", 0, 1, 32000, 13),
    ("base_10s", "base", 0, 0, 160000, 65),
    ("base_vadtrunc", "base", 0, 2, 159744, 12),
]


def build_hf(cfg, w):
    from transformers import MoonshineConfig, MoonshineForConditionalGeneration

    hcfg = MoonshineConfig(
        vocab_size=cfg.vocab,
        hidden_size=cfg.hidden,
        intermediate_size=cfg.ffn,
        encoder_num_hidden_layers=cfg.enc_layers,
        decoder_num_hidden_layers=cfg.dec_layers,
        encoder_num_attention_heads=cfg.heads,
        decoder_num_attention_heads=cfg.heads,
    )
    hcfg._attn_implementation = "eager"
    m = MoonshineForConditionalGeneration(hcfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in k or "inv_freq" in k for k in missing), missing
    return m


@torch.no_grad()
def run_case(name, arch, seed, clip, n, steps):
    cfg = ARCHS[arch]
    w = make_weights(cfg, seed)
    m = build_hf(cfg, w)
    audio = make_audio(clip, n)
    x = torch.from_numpy(audio)[None]
    # conv stem output (hf:573-577)
    enc_mod = m.model.encoder
    h = torch.tanh(enc_mod.conv1(x.unsqueeze(1)))
    h = enc_mod.groupnorm(h)
    h = torch.nn.functional.gelu(enc_mod.conv2(h))
    h = torch.nn.functional.gelu(enc_mod.conv3(h)).permute(0, 2, 1)[0]
    enc_out = enc_mod(x)
    enc = enc_out.last_hidden_state
    T = enc.shape[1]
    rows = np.unique(np.concatenate([np.arange(0, T, 7), [T - 1]]))
    from transformers.cache_utils import DynamicCache, EncoderDecoderCache

    past = EncoderDecoderCache(DynamicCache(config=m.config), DynamicCache(config=m.config))
    tokens = [cfg.bos]
    idxs, vals = [], []
    cur = torch.tensor([[cfg.bos]])
    for _ in range(steps):
        out = m(decoder_input_ids=cur, encoder_outputs=enc_out, past_key_values=past, use_cache=True)
        past = out.past_key_values
        lg = out.logits[0, -1].numpy()
        nxt = int(np.argmax(lg))
        top = np.argsort(-lg, kind="stable")[:8]
        idxs.append(top)
        vals.append(lg[top])
        tokens.append(nxt)
        cur = torch.tensor([[nxt]])
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"golden_{name}.npz")
    np.savez_compressed(
        out_path,
        arch=arch,
        seed=seed,
        clip=clip,
        n_samples=n,
        enc_rows=rows.astype(np.int32),
        enc=enc[0].numpy()[rows].astype(np.float32),
        enc_absmean=np.float32(enc.abs().mean().item()),
        conv3=h.numpy()[rows].astype(np.float32),
        tokens=np.asarray(tokens, np.int32),
        logit_idx=np.stack(idxs).astype(np.int32),
        logit_val=np.stack(vals).astype(np.float32),
    )
    print(name, "T=", T, "tokens", tokens[:12], "size", os.path.getsize(out_path))


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:] or None
    for c in CASES:
        if only and c[0] not in only:
            continue
        run_case(*c)
