"""Run the reference's own known-answer clips through the MI355X engine with the REAL Moonshine weights.

Needs network access (the build container and the GPU box have none, so this has NOT been run there):

    python tools/verify_real_checkpoint.py --arch base      # or tiny

Steps:
  1. download `UsefulSensors/moonshine-{arch}` from the HuggingFace hub: model.safetensors (loaded UNCHANGED: the engine's
     native format is the HF state_dict, tied proj_out included or absent) and tokenizer.json;
  2. write tokenizer.bin from tokenizer.json by the rule of the reference's scripts/convert_tokenizer.py:84-121
     (token id -> UTF-8 bytes of the piece, added_tokens override; length-prefixed entries);
  3. transcribe tests/golden/beckett.wav (reference fixture test-assets/beckett.wav) and, if present next to it,
     two_cities_16k.wav through the public C API (moonshine_load_transcriber_from_files /
     moonshine_transcribe_without_streaming, vad_threshold = 0) and expect the substrings the reference's own tests expect:
     "fail" (python/tests/test_modules.py:61-69), "best of times" / "worst of times" (TranscriberTest.java:124-125);
  4. with --hf, also run HuggingFace MoonshineForConditionalGeneration on the CPU and report whether the greedy ids agree.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tokenizer_bin_from_json(path_json: str, path_bin: str) -> int:
    from moonshine_amd.synth import encode_tokenizer_bin

    data = json.load(open(path_json, encoding="utf-8"))
    vocab = data["model"]["vocab"]
    tokens = [b""] * len(vocab)
    for piece, i in vocab.items():
        if i < len(tokens):
            tokens[i] = piece.encode("utf-8")
    for added in data.get("added_tokens", []):
        i, content = added.get("id"), added.get("content", "")
        if i is None:
            continue
        if i >= len(tokens):
            tokens.extend([b""] * (i - len(tokens) + 1))
        tokens[i] = content.encode("utf-8")
    with open(path_bin, "wb") as f:
        f.write(encode_tokenizer_bin(tokens))
    return len(tokens)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="base", choices=["tiny", "base"])
    ap.add_argument("--dir", default=None, help="model directory to (re)use; default ./real_models/<arch>")
    ap.add_argument("--hf", action="store_true", help="also compare the greedy ids with HuggingFace on the CPU")
    args = ap.parse_args()
    d = args.dir or os.path.join(ROOT, "real_models", args.arch)
    os.makedirs(d, exist_ok=True)
    if not os.path.exists(os.path.join(d, "model.safetensors")):
        from huggingface_hub import hf_hub_download

        for name in ("model.safetensors", "tokenizer.json", "config.json"):
            p = hf_hub_download(f"UsefulSensors/moonshine-{args.arch}", name)
            dst = os.path.join(d, name)
            if not os.path.exists(dst):
                os.symlink(p, dst)
    if not os.path.exists(os.path.join(d, "tokenizer.bin")):
        print("tokenizer.bin:", tokenizer_bin_from_json(os.path.join(d, "tokenizer.json"), os.path.join(d, "tokenizer.bin")), "entries")

    from moonshine_amd import api
    from moonshine_amd.hip_api import load_library
    import ctypes as C

    lib = load_library()

    def wav(path):
        r = C.c_int32(0)
        n = lib.msh_host_load_wav(path.encode(), None, 0, C.addressof(r))
        assert n > 0, path
        a = np.zeros(n, np.float32)
        lib.msh_host_load_wav(path.encode(), a.ctypes.data, n, C.addressof(r))
        return a, r.value

    t = api.Transcriber(d, api.ARCH_BASE if args.arch == "base" else api.ARCH_TINY,
                        {"vad_threshold": "0", "vad_max_segment_duration": "100000"})
    ok = True
    cases = [(os.path.join(ROOT, "tests", "golden", "beckett.wav"), ["fail"]),
             (os.path.join(ROOT, "tests", "golden", "two_cities_16k.wav"), ["best of times", "worst of times"])]
    for path, needles in cases:
        if not os.path.exists(path):
            continue
        audio, rate = wav(path)
        lines = t.transcribe_without_streaming(audio, sample_rate=rate)
        text = " ".join((l.text or "") for l in lines)
        hit = all(n in text.lower() for n in needles)
        ok &= hit
        print(("PASS " if hit else "FAIL ") + os.path.basename(path) + ": " + text)
        if args.hf:
            import torch
            from transformers import MoonshineForConditionalGeneration

            m = MoonshineForConditionalGeneration.from_pretrained(f"UsefulSensors/moonshine-{args.arch}").eval()
            seg = torch.from_numpy(audio[: (len(audio) // 512) * 512])[None]   # what the VAD hands the model (whole hops)
            with torch.no_grad():
                ids = m.generate(seg, max_new_tokens=int(np.ceil(seg.shape[1] / 16000 * 6.5)), do_sample=False)[0].tolist()
            print("  HF ids:", ids[:24], "...")
    t.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
