"""Diagnostic (GPU): per-clip error statistics of the long-clip parity cases (tests/test_gpu_long_parity.py): encoder
rel-RMS / max-abs and, per step range, the max-abs and RMS of (GPU logits - oracle logits) under teacher forcing."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moonshine_amd.hip_api import Engine  # noqa: E402
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors  # noqa: E402
from oracle import moonshine_ref as ref  # noqa: E402


def run(arch, seed, lens, picked, steps, base_seed):
    cfg = ARCHS[arch]
    w = make_weights(cfg, seed)
    e = Engine(0)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.safetensors")
        save_safetensors(p, w, {"arch": cfg.name, "heads": str(cfg.heads)})
        e.load_weights_file(p)
    clips = [make_audio(base_seed + i, n) for i, n in enumerate(lens)]
    e.set_keep_encoder_output(True)
    e.encode(clips)
    encs = {b: e.encoder_output(b) for b in picked}
    toks, _ = e.decode(forced_steps=steps)
    e.encode(clips)
    _, logits = e.decode(forced_steps=steps, teacher=np.asarray(toks, np.int32), want_logits=steps)
    for b in picked:
        enc = ref.encoder_forward(w, cfg, clips[b])
        err = encs[b] - enc
        o_toks, o_lg = ref.greedy_decode(w, cfg, enc, steps, ignore_eos=True, return_logits=True, teacher=toks[b])
        d = logits[:, b, :] - o_lg
        mx = np.abs(d).max(axis=1)
        rms = np.sqrt((d ** 2).mean(axis=1))
        print(f"{arch} clip {b} n={lens[b]} T={enc.shape[0]}: enc rel-RMS {np.sqrt((err**2).mean())/np.sqrt((enc**2).mean()):.2e} max {np.abs(err).max():.2e} | "
              f"logits max-abs: all {mx.max():.4f} first10 {mx[:10].max():.4f} last10 {mx[-10:].max():.4f} | rms mean {rms.mean():.4f} first10 {rms[:10].mean():.4f} last10 {rms[-10:].mean():.4f} | logit std {o_lg.std():.2f}", flush=True)


if __name__ == "__main__":
    run("tiny", 5, [480_000, 1_240_000, 160_000, 709_986, 48_000, 895, 1_000_003, 333_333], [0, 1, 2, 3, 4, 5], 100, 700)
    run("base", 0, [480_000, 160_000, 709_986, 900_000, 52_000, 1_240_000, 250_000, 20_000], [0, 1, 2, 4, 5, 7], 100, 800)
