#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_long_parity.py -q -x 2>&1 | tail -5
timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api > gpurun_out/r3y_bench.json 2> gpurun_out/r3y_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3y_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "serial", d["config"].get("serial_steps_value"), "fp8", d["config"].get("fp8_kv_value"))
for k in d["kernels"]:
    if k["kernel"].startswith(("enc_", "conv", "cross_kv")): print("  %-22s %.4f ms x %d  frac %.3f" % (k["kernel"], k["ms_per_launch"], k["launches_per_step"], k["frac"]))
PY
