"""tools/ort_to_safetensors.py (`.ort` initializer extractor with int8 per-channel dequantisation,
reference docs/models/quantization.md:3-7) on `.ort` files written HERE by a small encoder of the ORT FlatBuffer schema
(InferenceSession -> Model -> Graph -> initializers[Tensor]); no shipped Moonshine `.ort` exists in the reference checkout,
so the schema slots and the quantisation naming are restated, not verified against a real file -- the tool says so."""
import json
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ort_to_safetensors as ots  # noqa: E402


def build_ort(tensors: dict) -> bytes:
    """Encode {name: ndarray} as an ORT-format file.  Layout: root offset | "ORTM" | session table | model table | graph
    table | initializers vector | tensor tables | strings / dims / raw data.  Every uoffset points forward."""
    buf = bytearray(8)
    buf[4:8] = b"ORTM"
    patches = []   # (position of a uoffset field, key of its target)
    placed = {}

    def align(n):
        while len(buf) % n:
            buf.append(0)

    def table(key, n_fields, scalars: dict, refs: dict):
        """vtable then table; scalars: slot -> (fmt, value); refs: slot -> target key."""
        align(8)
        slots = sorted(list(scalars) + list(refs))
        # table body: soffset + fields in slot order, 8-byte fields first for alignment simplicity
        body_fields = []
        off = 4
        layout = {}
        for s_ in slots:
            size = 8 if (s_ in scalars and scalars[s_][0] == "q") else 4
            if size == 8 and off % 8 != 4 and (off % 8) != 0:
                pass
            layout[s_] = (off, size)
            off += size
        # make 8-byte scalars 8-aligned relative to the table start (which is made 8-aligned + 4 below)
        vt_size = 4 + 2 * n_fields
        vt_pos = len(buf)
        buf.extend(struct.pack("<HH", vt_size, off))
        for s_ in range(n_fields):
            buf.extend(struct.pack("<H", layout[s_][0] if s_ in layout else 0))
        align(4)
        tpos = len(buf)
        placed[key] = tpos
        buf.extend(struct.pack("<i", tpos - vt_pos))
        for s_ in slots:
            fpos = tpos + layout[s_][0]
            assert fpos == len(buf)
            if s_ in scalars:
                fmt, v = scalars[s_]
                buf.extend(struct.pack("<" + fmt, v))
            else:
                patches.append((fpos, refs[s_]))
                buf.extend(b"\0\0\0\0")
        return tpos

    names = list(tensors)
    table("session", 4, {}, {0: "ver", 1: "model"})
    table("model", 10, {0: ("q", 9)}, {7: "graph"})
    table("graph", 9, {}, {0: "inits"})
    align(4)
    placed["inits"] = len(buf)
    buf.extend(struct.pack("<I", len(names)))
    for i in range(len(names)):
        patches.append((len(buf), f"t{i}"))
        buf.extend(b"\0\0\0\0")
    code = {np.dtype(np.float32): 1, np.dtype(np.uint8): 2, np.dtype(np.int8): 3, np.dtype(np.int64): 7, np.dtype(np.float16): 10}
    for i, n in enumerate(names):
        a = tensors[n]
        table(f"t{i}", 7, {3: ("i", code[a.dtype])}, {0: f"n{i}", 2: f"d{i}", 4: f"r{i}"})
    align(4)
    placed["ver"] = len(buf)
    buf.extend(struct.pack("<I", 6) + b"1.23.2\0")
    for i, n in enumerate(names):
        a = np.ascontiguousarray(tensors[n])
        align(4)
        placed[f"n{i}"] = len(buf)
        raw = n.encode()
        buf.extend(struct.pack("<I", len(raw)) + raw + b"\0")
        align(4)
        while (len(buf) + 4) % 8:
            buf.append(0)
        placed[f"d{i}"] = len(buf)
        buf.extend(struct.pack("<I", a.ndim) + struct.pack(f"<{a.ndim}q", *a.shape))
        align(4)
        placed[f"r{i}"] = len(buf)
        rb = a.tobytes()
        buf.extend(struct.pack("<I", len(rb)) + rb)
    for fpos, key in patches:
        struct.pack_into("<I", buf, fpos, placed[key] - fpos)
    struct.pack_into("<I", buf, 0, placed["session"])
    return bytes(buf)


def test_reader_round_trip_and_per_channel_dequantisation(tmp_path):
    rng = np.random.default_rng(0)
    w_fp = rng.standard_normal((24, 10)).astype(np.float32) * np.linspace(0.1, 3.0, 10, dtype=np.float32)   # [K, N]: MatMul weight
    scale = (np.abs(w_fp).max(axis=0) / 127.0).astype(np.float32)                                          # one per OUTPUT channel
    zp = np.zeros(10, np.int8)
    q = np.clip(np.round(w_fp / scale), -127, 127).astype(np.int8)
    conv = rng.standard_normal((6, 3, 5)).astype(np.float32)
    cscale = (np.abs(conv).reshape(6, -1).max(axis=1) / 127.0).astype(np.float32)
    cq = np.clip(np.round(conv / cscale[:, None, None]), -127, 127).astype(np.int8)
    uq = rng.integers(0, 256, (8, 4)).astype(np.uint8)
    tensors = {
        "model.decoder.layers.0.mlp.fc2.weight_quantized": q,
        "model.decoder.layers.0.mlp.fc2.weight_scale": scale,
        "model.decoder.layers.0.mlp.fc2.weight_zero_point": zp,
        "frontend.conv1.weight_quantized": cq,
        "frontend.conv1.weight_scale": cscale,
        "frontend.conv1.weight_zero_point": np.zeros(6, np.int8),
        "per_tensor_quantized": uq,
        "per_tensor_scale": np.asarray([0.02], np.float32),
        "per_tensor_zero_point": np.asarray([128], np.uint8),
        "model.decoder.norm.weight": rng.standard_normal(16).astype(np.float32),
        "some_shape_constant": np.asarray([1, -1, 52], np.int64),
        "half": rng.standard_normal((3, 5)).astype(np.float16),
    }
    path = tmp_path / "decoder_model_merged.ort"
    path.write_bytes(build_ort(tensors))
    got = ots.read_initializers(str(path))
    assert list(got) == list(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        np.testing.assert_array_equal(got[k], v)
    deq = ots.dequantize(got)
    w = deq["model.decoder.layers.0.mlp.fc2.weight"]
    np.testing.assert_allclose(w, q.astype(np.float32) * scale[None, :], rtol=0, atol=0)
    assert np.abs(w - w_fp).max() <= scale.max() * 0.5 + 1e-7          # within half a quantisation step, per channel
    np.testing.assert_allclose(deq["frontend.conv1.weight"], cq.astype(np.float32) * cscale[:, None, None], rtol=0, atol=0)
    np.testing.assert_allclose(deq["per_tensor"], (uq.astype(np.float32) - 128.0) * 0.02, rtol=0, atol=1e-7)
    assert "model.decoder.norm.weight" in deq and "model.decoder.layers.0.mlp.fc2.weight_scale" not in deq


def test_cli_writes_safetensors_the_engine_reader_accepts(tmp_path):
    from moonshine_amd.synth import load_safetensors

    rng = np.random.default_rng(1)
    w_kn = rng.standard_normal((16, 8)).astype(np.float32)
    scale = (np.abs(w_kn).max(axis=0) / 127.0).astype(np.float32)
    tensors = {"onnx::MatMul_123_quantized": np.clip(np.round(w_kn / scale), -127, 127).astype(np.int8),
               "onnx::MatMul_123_scale": scale, "onnx::MatMul_123_zero_point": np.zeros(8, np.int8),
               "model.encoder.layer_norm.weight": rng.standard_normal(8).astype(np.float32)}
    ort = tmp_path / "encoder_model.ort"
    ort.write_bytes(build_ort(tensors))
    mp = tmp_path / "map.json"
    mp.write_text(json.dumps({"onnx::MatMul_123": {"name": "model.encoder.layers.0.self_attn.q_proj.weight", "transpose": True},
                              "model.encoder.layer_norm.weight": "model.encoder.layer_norm.weight"}))
    out = tmp_path / "model.safetensors"
    tool = os.path.join(ROOT, "tools", "ort_to_safetensors.py")
    r = subprocess.run([sys.executable, tool, str(ort), str(out), "--map", str(mp)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got, meta = load_safetensors(str(out))
    assert set(got) == {"model.encoder.layers.0.self_attn.q_proj.weight", "model.encoder.layer_norm.weight"}
    assert got["model.encoder.layers.0.self_attn.q_proj.weight"].shape == (8, 16)        # [N, K]: the HF nn.Linear layout
    np.testing.assert_allclose(got["model.encoder.layers.0.self_attn.q_proj.weight"],
                               (tensors["onnx::MatMul_123_quantized"].astype(np.float32) * scale).T)
    listing = subprocess.run([sys.executable, tool, str(ort), "--list"], capture_output=True, text=True)
    assert "onnx::MatMul_123_quantized" in listing.stdout and "int8" in listing.stdout
    bad = tmp_path / "bad.ort"
    bad.write_bytes(b"\0" * 64)
    assert subprocess.run([sys.executable, tool, str(bad), "--list"], capture_output=True, text=True).returncode != 0
