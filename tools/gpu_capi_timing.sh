#!/bin/bash
# C-API batch call: tests + phase times of the call under the reference's default options (MSH_HOST_TIMING)
set -u
tag=${1:-capi}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_capi.py tests/test_gpu_capi_threads.py -q -x 2>&1 | tail -2
MSH_HOST_TIMING=1 timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-fp8 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "c_api", d["config"].get("c_api_batch_value"), "default_vad", d["config"].get("c_api_batch_default_vad_value"))
PY
grep "batch call" gpurun_out/${tag}_bench.err | tail -8
