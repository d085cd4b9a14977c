#!/bin/bash
# C-API batch call with kernel_set=per_call against auto (= uniform with batch_clips >= 192), alternating on one box, and the
# single-clip p50 with and without msh_set_uniform_kernels.
export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do
for KS in per_call auto; do
  timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-fp8 --capi-kernel-set $KS > gpurun_out/ks.json 2> gpurun_out/ks.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ks.json").read().strip().splitlines()[-1])
c = d["c_api_batch"]
print("kernel_set=$KS rep $rep: c_api", c["value"], c["ms_per_call"], "default_vad", c["default_vad"]["value"], c["default_vad"]["ms_per_call"], "lines", c["default_vad"]["lines"], "pcm16", c["default_vad"]["pcm16"]["value"])
PY
done
done
cat > /tmp/lat.py <<'PY'
import os, sys, time, tempfile, statistics
sys.path.insert(0, ".")
import numpy as np
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors
cfg = ARCHS["base"]; w = make_weights(cfg, 0)
d = tempfile.mkdtemp(); path = os.path.join(d, "m.safetensors"); save_safetensors(path, w, {"arch": cfg.name, "heads": str(cfg.heads)})
e = Engine(0); e.load_weights_file(path)
clip = [make_audio(1, 160000)]
for uni in (False, True):
    e.set_uniform_kernels(uni)
    for _ in range(3): e.transcribe_tokens(clip, forced_steps=65)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); e.transcribe_tokens(clip, forced_steps=65); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"one 10 s clip, uniform_kernels={uni}: p50 {statistics.median(ts):.2f} ms")
PY
python /tmp/lat.py
