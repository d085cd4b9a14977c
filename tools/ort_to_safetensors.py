"""Pull the weights out of an ONNX Runtime `.ort` model file (what the reference ships: encoder_model.ort,
decoder_model_merged.ort, frontend / encoder / adapter / cross_kv / decoder_kv .ort) into a safetensors file, undoing the
int8 weight quantisation on the way (reference docs/models/quantization.md:3-7: eight-bit weights with ONE SCALE PER OUTPUT
CHANNEL; scripts/quantize-streaming-model.sh).

    python tools/ort_to_safetensors.py decoder_model_merged.ort out.safetensors [--map names.json] [--list]

What this is and is not.  An `.ort` file is a FlatBuffer (file identifier "ORTM") whose root `InferenceSession` holds
`model.graph.initializers`: a vector of `Tensor {name, dims, data_type, raw_data}`.  The reader below walks exactly that
path with a hand-written FlatBuffer decoder (no onnxruntime / flatbuffers package needed); the table field order is
restated from onnxruntime's `core/flatbuffers/schema/ort.fbs` (ORT 1.23, the version the reference pins).  It has been
exercised on `.ort` files written by the test-suite's own encoder of that schema (tests/test_ort_extract.py) -- NOT on a
shipped Moonshine `.ort`, none of which is in the reference checkout, and there is no network here.  Two things are
therefore stated, not verified: (1) the initializer names of the shipped graphs -- the tool keeps whatever names the file
holds, and `--map` renames them to the HuggingFace names the engine loads (a JSON object {"ort name": "hf name"} or
{"ort name": {"name": "hf name", "transpose": true}} for MatMul weights stored [K, N]); (2) the quantisation naming: the
ONNX convention `<w>_quantized` (int8 / uint8) + `<w>_scale` + `<w>_zero_point`, per-tensor or per-channel, is what is
dequantised ((q - zero_point) * scale along the axis whose length equals the scale's); anything else is copied as is.
"""
from __future__ import annotations

import argparse
import json
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# ONNX TensorProto.DataType -> numpy
DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64,
          12: np.uint32, 13: np.uint64}
BF16 = 16


class FlatBuffer:
    """Just enough of the FlatBuffers wire format to read tables, strings and vectors."""

    def __init__(self, data: bytes):
        self.b = data

    def u16(self, p): return struct.unpack_from("<H", self.b, p)[0]
    def u32(self, p): return struct.unpack_from("<I", self.b, p)[0]
    def i32(self, p): return struct.unpack_from("<i", self.b, p)[0]
    def i64(self, p): return struct.unpack_from("<q", self.b, p)[0]

    def root(self) -> int:
        return self.u32(0)

    def field(self, table: int, index: int) -> int | None:
        """Absolute position of field `index` of the table at `table`, or None when absent (default value)."""
        vt = table - self.i32(table)
        vsize = self.u16(vt)
        slot = 4 + 2 * index
        if slot >= vsize:
            return None
        off = self.u16(vt + slot)
        return table + off if off else None

    def indirect(self, p: int) -> int:
        return p + self.u32(p)

    def table(self, table: int, index: int) -> int | None:
        p = self.field(table, index)
        return None if p is None else self.indirect(p)

    def string(self, table: int, index: int) -> str | None:
        p = self.field(table, index)
        if p is None:
            return None
        s = self.indirect(p)
        n = self.u32(s)
        return self.b[s + 4:s + 4 + n].decode("utf-8", errors="replace")

    def vector(self, table: int, index: int) -> tuple[int, int]:
        """(position of element 0, length) of a vector field; (0, 0) when absent."""
        p = self.field(table, index)
        if p is None:
            return 0, 0
        v = self.indirect(p)
        return v + 4, self.u32(v)


# vtable slots (declaration order in ort.fbs)
SESSION_MODEL = 1
MODEL_GRAPH = 7
GRAPH_INITIALIZERS = 0
T_NAME, T_DIMS, T_DTYPE, T_RAW, T_EXTERNAL = 0, 2, 3, 4, 6


def read_initializers(path: str) -> dict[str, np.ndarray]:
    data = open(path, "rb").read()
    if len(data) < 12 or data[4:8] != b"ORTM":
        raise SystemExit(f"{path}: not an ORT format model (file identifier {data[4:8]!r}, expected b'ORTM')")
    fb = FlatBuffer(data)
    session = fb.root()
    model = fb.table(session, SESSION_MODEL)
    graph = fb.table(model, MODEL_GRAPH) if model is not None else None
    if graph is None:
        raise SystemExit(f"{path}: no model.graph in the session")
    pos, n = fb.vector(graph, GRAPH_INITIALIZERS)
    out: dict[str, np.ndarray] = {}
    for i in range(n):
        t = fb.indirect(pos + 4 * i)
        name = fb.string(t, T_NAME) or f"initializer_{i}"
        dpos, dn = fb.vector(t, T_DIMS)
        dims = [fb.i64(dpos + 8 * k) for k in range(dn)]
        p = fb.field(t, T_DTYPE)
        dtype = fb.i32(p) if p is not None else 0
        rpos, rn = fb.vector(t, T_RAW)
        ext = fb.field(t, T_EXTERNAL)
        if rn == 0 and ext is not None and fb.i64(ext) >= 0:
            raise SystemExit(f"{path}: tensor '{name}' keeps its data in an external file (offset {fb.i64(ext)}): not supported")
        raw = data[rpos:rpos + rn]
        if dtype == BF16:
            a = (np.frombuffer(raw, np.uint16).astype(np.uint32) << 16).view(np.float32)
        elif dtype in DTYPES:
            a = np.frombuffer(raw, DTYPES[dtype])
        else:
            print(f"skipping '{name}': data type {dtype} not handled", file=sys.stderr)
            continue
        want = int(np.prod(dims)) if dims else 1
        if a.size != want:
            raise SystemExit(f"{path}: tensor '{name}' has {a.size} elements for dims {dims}")
        out[name] = a.reshape(dims) if dims else a.reshape(())
    return out


def dequantize(tensors: dict[str, np.ndarray]) -> dict[str, np.ndarray]:
    """ONNX convention: <w>_quantized + <w>_scale (+ <w>_zero_point); per-tensor or per-channel."""
    out: dict[str, np.ndarray] = {}
    used = set()
    for name, q in tensors.items():
        if not name.endswith("_quantized") or q.dtype not in (np.int8, np.uint8):
            continue
        base = name[: -len("_quantized")]
        scale = tensors.get(base + "_scale")
        if scale is None:
            continue
        zp = tensors.get(base + "_zero_point")
        scale = np.asarray(scale, np.float32).reshape(-1)
        zpv = np.zeros_like(scale, dtype=np.float32) if zp is None else np.asarray(zp, np.float32).reshape(-1)
        if zpv.size == 1 and scale.size > 1:
            zpv = np.full(scale.size, zpv[0], np.float32)
        x = q.astype(np.float32)
        if scale.size == 1:
            w = (x - zpv[0]) * scale[0]
        else:
            axes = [ax for ax, d in enumerate(q.shape) if d == scale.size]
            if not axes:
                raise SystemExit(f"'{name}': scale of length {scale.size} matches no axis of {q.shape}")
            # one scale per OUTPUT channel: the last axis of a MatMul weight [K, N], axis 0 of a Conv weight [Cout, ...]
            ax = axes[-1] if q.ndim == 2 else axes[0]
            shape = [1] * q.ndim
            shape[ax] = scale.size
            w = (x - zpv.reshape(shape)) * scale.reshape(shape)
        out[base] = w.astype(np.float32)
        used.update({name, base + "_scale", base + "_zero_point"})
    for name, a in tensors.items():
        if name not in used and name not in out:
            out[name] = a
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ort")
    ap.add_argument("out", nargs="?")
    ap.add_argument("--map", help="JSON: ort initializer name -> HF name (or {name, transpose})")
    ap.add_argument("--list", action="store_true", help="print name / dtype / shape of every initializer and stop")
    args = ap.parse_args()
    raw = read_initializers(args.ort)
    if args.list or not args.out:
        for k, v in raw.items():
            print(f"{k:60s} {str(v.dtype):8s} {list(v.shape)}")
        return
    tensors = dequantize(raw)
    if args.map:
        m = json.load(open(args.map))
        renamed = {}
        for k, v in tensors.items():
            e = m.get(k)
            if e is None:
                continue
            if isinstance(e, str):
                e = {"name": e}
            renamed[e["name"]] = np.ascontiguousarray(v.T if e.get("transpose") else v)
        tensors = renamed
    from moonshine_amd.synth import save_safetensors

    keep = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in tensors.items() if v.dtype.kind == "f" and v.ndim >= 1}
    save_safetensors(args.out, keep, {"source": os.path.basename(args.ort)})
    print(f"wrote {len(keep)} tensors to {args.out}")


if __name__ == "__main__":
    main()
