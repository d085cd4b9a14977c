#!/bin/bash
# GroupNorm statistics A/B (MSH_GN_ROWSUMS=0 pass over the conv1 output / 1 row sums out of conv1's epilogue): parity file, then
# headline + conv1 / statistics times alternating on one box.
set -u
export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
for rep in 1 2; do
for KO in 0 1; do
  MSH_GN_ROWSUMS=$KO timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
ks = {k["kernel"]: k for k in d["kernels"]}
print("ROWSUMS=$KO rep $rep:", d["value"], "serial", d["serial_steps"]["value"], "ids", d["config"]["ids_match_serial_pass"],
      "conv1 ms", ks["conv1_tanh_gemm"]["ms_per_launch"], "gn ms", ks["groupnorm_stats"]["ms_per_launch"], "conv2", ks["conv2_gelu_gemm"]["ms_per_launch"])
PY
done
done
