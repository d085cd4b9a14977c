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

Two further modes, each a closed loop that exits non-zero on ANY mismatch and writes a JSON report (--report, default
profiles/verify_<mode>_report.json) that can be committed as evidence:

    python tools/verify_real_checkpoint.py --silero      # pins oracle / host / device Silero VAD on the PUBLISHED model
    python tools/verify_real_checkpoint.py --ort         # the reference's shipped int8 .ort files through the extractor

  --silero  downloads the upstream Silero VAD `.onnx` at the commit and SHA-256 the reference pins
            (/root/reference/scripts/generate-silero-vad-data.py:60-68), pulls the 16 kHz branch's tensors out of it
            (needs `onnx`), writes silero_vad.safetensors, runs the `.onnx` in onnxruntime the way the reference drives it
            (core/silero-vad.cpp:78-173: 64 samples of context + 512-sample hop, state [2,1,128], sr = 16000) over the first
            100 hops of beckett.wav, and requires: oracle/silero_ref.py within 1e-5 of onnxruntime, the library's host
            network within 1e-4, the device network (when a GPU is visible) within 1e-4.
  --ort     downloads the reference's `base-en` components (core/moonshine-model-catalog.cpp:36,86-113:
            https://download.moonshine.ai/model/base-en/quantized/base-en/{encoder_model.ort, decoder_model_merged.ort,
            tokenizer.bin}), lists / dequantises their initializers (tools/ort_to_safetensors.py), matches every
            dequantised matrix to a tensor of the HuggingFace float checkpoint (same shape or its transpose, smallest
            relative error), requires every float weight of the HF state_dict to be matched within 3e-2 relative RMS (int8,
            one scale per output channel: docs/models/quantization.md:3-7), writes the matched tensors under HF names and
            transcribes the two fixtures with THEM, expecting the same substrings.
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


SILERO_VAD_URL = ("https://raw.githubusercontent.com/snakers4/silero-vad/"
                  "b163605b3f44c3aadf28f97b125a2f7c461e9a7f/src/silero_vad/data/silero_vad.onnx")
SILERO_VAD_SHA256 = "1a153a22f4509e292a94e67d6f9b85e8deb25b4988682b7e174c65279d8788e3"
CDN_BASE_EN = "https://download.moonshine.ai/model/base-en/quantized/base-en"
FIXTURES = [("beckett.wav", ["fail"]), ("two_cities_16k.wav", ["best of times", "worst of times"])]


def _download(url: str, dst: str, sha256: str | None = None) -> str:
    import hashlib
    import urllib.request

    if not os.path.exists(dst):
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with urllib.request.urlopen(url) as r:
            data = r.read()
        with open(dst, "wb") as f:
            f.write(data)
    if sha256 is not None:
        got = hashlib.sha256(open(dst, "rb").read()).hexdigest()
        if got != sha256:
            raise SystemExit(f"{dst}: SHA-256 {got}, expected {sha256}")
    return dst


def _load_wav(lib, path):
    import ctypes as C

    r = C.c_int32(0)
    n = lib.msh_host_load_wav(path.encode(), None, 0, C.addressof(r))
    assert n > 0, path
    a = np.zeros(n, np.float32)
    lib.msh_host_load_wav(path.encode(), a.ctypes.data, n, C.addressof(r))
    return a, r.value


def _write_report(args, mode: str, report: dict, ok: bool) -> None:
    report["ok"] = bool(ok)
    path = args.report or os.path.join(ROOT, "profiles", f"verify_{mode}_report.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(report, f, indent=1)
    print(("PASS" if ok else "FAIL"), mode, "->", path)
    sys.exit(0 if ok else 1)


def verify_silero(args) -> None:
    import ctypes as C

    import onnx
    import onnxruntime as ort
    from onnx import numpy_helper

    from moonshine_amd.hip_api import load_library
    from moonshine_amd.synth import save_safetensors
    from oracle.silero_ref import SileroRef
    from tools.convert_silero_vad import WANT

    d = args.dir or os.path.join(ROOT, "real_models", "silero")
    src = _download(SILERO_VAD_URL, os.path.join(d, "silero_vad.onnx"), SILERO_VAD_SHA256)
    report = {"mode": "silero", "source": SILERO_VAD_URL, "sha256": SILERO_VAD_SHA256, "checks": []}
    # every initializer and Constant of the graph and of its If / Loop sub-graphs (the model branches on the sample rate)
    found: dict[str, np.ndarray] = {}

    def walk(g):
        for t in g.initializer:
            found[t.name] = numpy_helper.to_array(t)
        for node in g.node:
            for a in node.attribute:
                if a.type == onnx.AttributeProto.GRAPH:
                    walk(a.g)
                elif a.type == onnx.AttributeProto.GRAPHS:
                    for sg in a.graphs:
                        walk(sg)
                elif a.type == onnx.AttributeProto.TENSOR and node.op_type == "Constant":
                    found[node.output[0]] = numpy_helper.to_array(a.t)

    walk(onnx.load(src).graph)
    out, problems = {}, []
    for name, shape in WANT.items():
        hits = [k for k, v in found.items() if tuple(v.shape) == shape and "8k" not in k and (k.endswith(name) or name.split(".")[-2] in k)]
        exact = [k for k in hits if k.endswith(name)]
        hits = exact or hits
        if len(hits) != 1:
            problems.append(f"{name}: {len(hits)} candidates {hits[:4]} among tensors of shape {shape}: "
                            f"{[k for k, v in found.items() if tuple(v.shape) == shape][:6]}")
            continue
        out[name] = np.ascontiguousarray(found[hits[0]], np.float32)
    report["tensor_mapping_problems"] = problems
    if problems:
        report["all_tensors"] = {k: list(v.shape) for k, v in found.items()}
        _write_report(args, "silero", report, False)
    st = os.path.join(d, "silero_vad.safetensors")
    save_safetensors(st, out, {"source": "silero-vad v5, 16 kHz branch", "sha256": SILERO_VAD_SHA256})
    lib = load_library()
    audio, rate = _load_wav(lib, os.path.join(ROOT, "tests", "golden", "beckett.wav"))
    assert rate == 16000
    hops = 100
    # the published model, driven as core/silero-vad.cpp:78-173 drives it
    sess = ort.InferenceSession(src, providers=["CPUExecutionProvider"])
    state, ctx, want = np.zeros((2, 1, 128), np.float32), np.zeros(64, np.float32), []
    for i in range(hops):
        x = np.concatenate([ctx, audio[i * 512:(i + 1) * 512]])[None].astype(np.float32)
        p, state = sess.run(None, {"input": x, "state": state, "sr": np.array(16000, np.int64)})
        ctx = x[0, -64:]
        want.append(float(np.asarray(p).reshape(-1)[0]))
    want = np.asarray(want, np.float32)
    net = SileroRef(out)
    ref = np.asarray([net.predict(audio[i * 512:(i + 1) * 512]) for i in range(hops)], np.float32)
    blob = open(st, "rb").read()
    host = np.zeros(hops, np.float32)
    lib.msh_host_silero_probabilities.restype = C.c_int64
    lib.msh_host_silero_probabilities.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    seg = np.ascontiguousarray(audio[: hops * 512])
    assert lib.msh_host_silero_probabilities(blob, len(blob), seg.ctypes.data, seg.shape[0], host.ctypes.data, hops, None) == hops
    checks = [("oracle/silero_ref.py vs onnxruntime", float(np.abs(ref - want).max()), 1e-5),
              ("libmoonshine host network vs onnxruntime", float(np.abs(host - want).max()), 1e-4)]
    if lib.msh_device_count() > 0:
        h = C.c_void_p()
        lib.msh_silero_create.restype = C.c_int32
        lib.msh_silero_create.argtypes = [C.c_int32, C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]
        lib.msh_silero_probabilities.restype = C.c_int64
        lib.msh_silero_probabilities.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint64, C.c_void_p, C.c_uint64]
        assert lib.msh_silero_create(0, blob, len(blob), C.byref(h)) == 0
        dev = np.zeros(hops, np.float32)
        ptrs, lens = (C.c_void_p * 1)(seg.ctypes.data), (C.c_uint64 * 1)(seg.shape[0])
        assert lib.msh_silero_probabilities(h, ptrs, lens, 1, dev.ctypes.data, hops) == hops
        checks.append(("libmoonshine device network vs onnxruntime", float(np.abs(dev - want).max()), 1e-4))
    else:
        report["device"] = "no GPU visible: device network not checked"
    ok = True
    for name, err, tol in checks:
        report["checks"].append({"what": name, "max_abs_diff": err, "tolerance": tol, "ok": err <= tol})
        print(f"{'PASS' if err <= tol else 'FAIL'} {name}: max |diff| {err:.3e} (tolerance {tol:g})")
        ok &= err <= tol
    report["probability_range"] = [float(want.min()), float(want.max())]
    _write_report(args, "silero", report, ok)


def verify_ort(args) -> None:
    from moonshine_amd import api
    from moonshine_amd.hip_api import load_library
    from moonshine_amd.synth import load_safetensors, save_safetensors
    from tools.ort_to_safetensors import dequantize, read_initializers

    d = args.dir or os.path.join(ROOT, "real_models", "base-en-ort")
    report = {"mode": "ort", "source": CDN_BASE_EN, "files": {}, "unmatched_hf_tensors": [], "matches": []}
    tensors: dict[str, np.ndarray] = {}
    for name in ("encoder_model.ort", "decoder_model_merged.ort", "tokenizer.bin"):
        p = _download(f"{CDN_BASE_EN}/{name}", os.path.join(d, name))
        report["files"][name] = os.path.getsize(p)
        if name.endswith(".ort"):
            raw = read_initializers(p)
            report["files"][name + ":initializers"] = len(raw)
            for k, v in dequantize(raw).items():
                if v.dtype.kind == "f" and v.ndim >= 1:
                    tensors[f"{name.split('_')[0]}:{k}"] = np.asarray(v, np.float32)
    from huggingface_hub import hf_hub_download

    hf, _ = load_safetensors(hf_hub_download("UsefulSensors/moonshine-base", "model.safetensors"))
    out, ok = {}, True
    by_shape: dict[tuple, list[str]] = {}
    for k, v in tensors.items():
        by_shape.setdefault(tuple(v.shape), []).append(k)
    for name, w in hf.items():
        w = np.asarray(w, np.float32)
        best = (None, np.inf, False)
        for shape, tr in ((tuple(w.shape), False), (tuple(w.shape[::-1]), True)) if w.ndim == 2 else ((tuple(w.shape), False),):
            for k in by_shape.get(shape, []):
                c = tensors[k].T if tr else tensors[k]
                err = float(np.sqrt(((c - w) ** 2).mean()) / max(np.sqrt((w ** 2).mean()), 1e-12))
                if err < best[1]:
                    best = (k, err, tr)
        tol = 3e-2 if w.ndim >= 2 else 1e-3
        good = best[0] is not None and best[1] <= tol
        report["matches"].append({"hf": name, "ort": best[0], "transposed": best[2], "rel_rms": None if best[0] is None else best[1], "ok": good})
        if good:
            out[name] = np.ascontiguousarray(tensors[best[0]].T if best[2] else tensors[best[0]])
        else:
            report["unmatched_hf_tensors"].append(name)
            ok = False
    print(f"{len(out)} of {len(hf)} HF tensors found in the .ort files within tolerance; unmatched: {report['unmatched_hf_tensors'][:8]}")
    if ok:
        md = os.path.join(d, "engine_dir")
        os.makedirs(md, exist_ok=True)
        save_safetensors(os.path.join(md, "model.safetensors"), out, {"source": "base-en .ort files, dequantised"})
        import shutil

        shutil.copyfile(os.path.join(d, "tokenizer.bin"), os.path.join(md, "tokenizer.bin"))
        lib = load_library()
        t = api.Transcriber(md, api.ARCH_BASE, {"vad_threshold": "0", "vad_max_segment_duration": "100000"})
        report["transcripts"] = {}
        for wav_name, needles in FIXTURES:
            path = os.path.join(ROOT, "tests", "golden", wav_name)
            audio, rate = _load_wav(lib, path)
            text = " ".join((l.text or "") for l in t.transcribe_without_streaming(audio, sample_rate=rate))
            hit = all(n in text.lower() for n in needles)
            report["transcripts"][wav_name] = {"text": text, "expected_substrings": needles, "ok": hit}
            print(("PASS " if hit else "FAIL ") + wav_name + ": " + text)
            ok &= hit
        t.close()
    _write_report(args, "ort", report, ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="base", choices=["tiny", "base"])
    ap.add_argument("--dir", default=None, help="model directory to (re)use; default ./real_models/<arch>")
    ap.add_argument("--hf", action="store_true", help="also compare the greedy ids with HuggingFace on the CPU")
    ap.add_argument("--silero", action="store_true", help="pin the Silero VAD implementations on the published model")
    ap.add_argument("--ort", action="store_true", help="the reference's shipped base-en .ort files through the extractor")
    ap.add_argument("--report", default=None, help="where to write the JSON report of --silero / --ort")
    args = ap.parse_args()
    if args.silero:
        return verify_silero(args)
    if args.ort:
        return verify_ort(args)
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
    from moonshine_amd.hip_api import ModelInfo, load_library
    import ctypes as C

    lib = load_library()
    # the loader's own verdict on the download, without a GPU: names, shapes, dtypes, tied head, nothing unknown
    blob = open(os.path.join(d, "model.safetensors"), "rb").read()
    lib.msh_host_check_weights.restype = C.c_int32
    lib.msh_host_check_weights.argtypes = [C.c_char_p, C.c_uint64, C.c_int32, C.POINTER(ModelInfo), C.c_char_p, C.c_uint64]
    mi, err = ModelInfo(), C.create_string_buffer(2048)
    if lib.msh_host_check_weights(blob, len(blob), 1 if args.arch == "base" else 0, C.byref(mi), err, len(err)) != 0:
        print("FAIL checkpoint refused by the loader:", err.value.decode(errors="replace"))
        sys.exit(1)
    print(f"checkpoint accepted: hidden {mi.hidden}, ffn {mi.ffn}, layers {mi.enc_layers}+{mi.dec_layers}, heads {mi.heads}, vocab {mi.vocab}")
    del blob

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
