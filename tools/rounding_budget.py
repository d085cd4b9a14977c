"""Where the logit error of the bf16 engine comes from, rounding point by rounding point (the review's question about the
4.3e-2 max-abs against a 5e-2 budget on fan-in-scaled random weights).

A numpy restatement of the decoder step with SWITCHABLE bf16 rounding at exactly the places the HIP kernels round
(the residual stream, q, softmax and every accumulation stay fp32):
    w      every GEMM weight (and the tied embedding as LM-head operand) is bf16
    enc    the encoder output is stored as bf16 (the cross K/V projection and the absorbed form read it)
    kv     the cross K/V and the self-attention cache are stored as bf16
    ln     LayerNorm outputs = the A operands of the QKV / cross-q / fc1 GEMMs and of the LM head are bf16
    attn   attention outputs (A operand of the o-proj GEMMs) are bf16
    mlp    the SwiGLU product (A operand of fc2) is bf16
Each point is switched on ALONE against the fp32 oracle, then all together; on a GPU box the engine's own encoder output and
logits are added (what the encoder's rounding alone contributes downstream, and the real total).

    python tools/rounding_budget.py [--clips 3] [--steps 24] [--arch base] [--gpu]
Writes one JSON object (also merged into profiles/parity_margins.json under "rounding_budget" by tests/margins.py's format).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moonshine_amd.synth import ARCHS, make_audio, make_weights  # noqa: E402
from oracle import moonshine_ref as ref  # noqa: E402

F32 = np.float32


def bf16(x):
    u = np.ascontiguousarray(x, F32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)).view(F32).reshape(np.shape(x))


class Rounded:
    """decoder_forward of oracle/moonshine_ref.py (hf modeling_moonshine.py:639-711) with the engine's rounding points."""

    def __init__(self, w, cfg, enc, on):
        self.cfg, self.on = cfg, on
        r = (lambda a: bf16(a)) if "w" in on else (lambda a: a)
        self.w = {k: (r(v) if v.ndim == 2 else v) for k, v in w.items() if k.startswith("model.decoder")}
        self.E_in = w["model.decoder.embed_tokens.weight"]            # the input lookup reads the fp32 copy
        H = cfg.heads
        e = bf16(enc) if "enc" in on else enc
        self.cross = []
        for l in range(cfg.dec_layers):
            p = f"model.decoder.layers.{l}.encoder_attn."
            k, v = e @ self.w[p + "k_proj.weight"].T, e @ self.w[p + "v_proj.weight"].T
            if "kv" in on:
                k, v = bf16(k), bf16(v)
            self.cross.append((ref._heads(k.astype(F32), H), ref._heads(v.astype(F32), H)))
        self.sk = [np.zeros((H, 0, cfg.head_dim), F32) for _ in range(cfg.dec_layers)]
        self.sv = [np.zeros((H, 0, cfg.head_dim), F32) for _ in range(cfg.dec_layers)]

    def step(self, tok):
        cfg, w, on, H = self.cfg, self.w, self.on, self.cfg.heads
        A = (lambda a: bf16(a)) if "ln" in on else (lambda a: a)
        h = self.E_in[[tok]].astype(F32)
        past = self.sk[0].shape[1]
        cos, sin = ref.rope_tables(cfg, np.arange(past, past + 1))
        for l in range(cfg.dec_layers):
            p = f"model.decoder.layers.{l}."
            y = A(ref.layer_norm_nobias(h, np.ones(cfg.hidden, F32))) * 1.0
            g = w[p + "input_layernorm.weight"]      # (the engine folds gamma into the weights: y g W^T = y (W diag g)^T)
            wq, wk, wv = (w[p + f"self_attn.{n}_proj.weight"] * g[None, :] for n in "qkv")
            if "w" in on:
                wq, wk, wv = bf16(wq), bf16(wk), bf16(wv)
            q = ref.apply_rope(ref._heads(y @ wq.T, H), cos, sin)
            k = ref.apply_rope(ref._heads(y @ wk.T, H), cos, sin)
            v = ref._heads(y @ wv.T, H)
            if "kv" in on:
                k, v = bf16(k), bf16(v)
            self.sk[l] = np.concatenate([self.sk[l], k], axis=1)
            self.sv[l] = np.concatenate([self.sv[l], v], axis=1)
            a = ref.attention(q, self.sk[l], self.sv[l], causal_offset=past)
            if "attn" in on:
                a = bf16(a)
            h = h + a @ w[p + "self_attn.o_proj.weight"].T
            y = A(ref.layer_norm_nobias(h, np.ones(cfg.hidden, F32)))
            wq = w[p + "encoder_attn.q_proj.weight"] * w[p + "post_attention_layernorm.weight"][None, :]
            if "w" in on:
                wq = bf16(wq)
            q = ref._heads(y @ wq.T, H)
            a = ref.attention(q, *self.cross[l])
            if "attn" in on:
                a = bf16(a)
            h = h + a @ w[p + "encoder_attn.o_proj.weight"].T
            y = A(ref.layer_norm_nobias(h, np.ones(cfg.hidden, F32)))
            w1 = w[p + "mlp.fc1.weight"] * w[p + "final_layernorm.weight"][None, :]
            if "w" in on:
                w1 = bf16(w1)
            y = y @ w1.T + w[p + "mlp.fc1.bias"]
            val, gate = np.split(y, 2, axis=-1)
            y = ref.silu(gate) * val
            if "mlp" in on:
                y = bf16(y)
            h = (h + y @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"]).astype(F32)
        y = ref.layer_norm_nobias(h, w["model.decoder.norm.weight"])
        if "ln" in on:
            y = bf16(y)
        return (y @ self.w["model.decoder.embed_tokens.weight"].T).astype(F32)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=3)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--arch", default="base")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cfg = ARCHS[a.arch]
    w = make_weights(cfg, 0)
    clips = [make_audio(900 + i, int(a.seconds * 16000)) for i in range(a.clips)]
    encs = [ref.encoder_forward(w, cfg, c) for c in clips]
    gold = [ref.greedy_decode(w, cfg, e, a.steps, ignore_eos=True, return_logits=True) for e in encs]
    points = ["w", "enc", "kv", "ln", "attn", "mlp"]
    configs = [("none (restatement check)", set())] + [(p, {p}) for p in points] + [("all", set(points))]
    res = {}
    for name, on in configs:
        worst = 0.0
        for enc, (toks, lg) in zip(encs, gold):
            m = Rounded(w, cfg, enc, on)
            for i in range(a.steps):
                worst = max(worst, float(np.abs(m.step(toks[i]) - lg[i]).max()))
        res[name] = worst
        print(f"{name:28s} logits max-abs vs fp32 oracle {worst:.3e}", flush=True)
    if a.gpu:
        from moonshine_amd.hip_api import Engine
        from moonshine_amd.synth import save_safetensors

        save_safetensors("/tmp/rb.safetensors", w, {"arch": cfg.name})
        e = Engine(0)
        e.load_weights_file("/tmp/rb.safetensors")
        e.set_keep_encoder_output(True)
        e.encode(clips)
        genc = [e.encoder_output(i) for i in range(a.clips)]
        teacher = np.asarray([t[: a.steps + 1] for t, _ in gold], np.int32)
        _, glog = e.decode(forced_steps=a.steps, teacher=teacher, want_logits=a.steps)
        worst_enc = worst_all = worst_gpu = 0.0
        for b, (enc, (toks, lg)) in enumerate(zip(genc, gold)):
            m0, m1 = Rounded(w, cfg, enc, set()), Rounded(w, cfg, enc, set(points))
            for i in range(a.steps):
                worst_enc = max(worst_enc, float(np.abs(m0.step(toks[i]) - lg[i]).max()))
                worst_all = max(worst_all, float(np.abs(m1.step(toks[i]) - lg[i]).max()))
                worst_gpu = max(worst_gpu, float(np.abs(glog[i, b] - lg[i]).max()))
        res["engine's encoder output, fp32 decoder"] = worst_enc
        res["engine's encoder output, all decoder points"] = worst_all
        res["engine (measured)"] = worst_gpu
        res["encoder_rel_rms"] = max(float(np.sqrt(((g - o) ** 2).mean()) / np.sqrt((o ** 2).mean())) for g, o in zip(genc, encs))
        for k in list(res)[-4:]:
            print(f"{k:46s} {res[k]:.3e}")
    out = {"arch": a.arch, "clips": a.clips, "steps": a.steps, "seconds": a.seconds, "logits_max_abs_vs_fp32_oracle": res}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
