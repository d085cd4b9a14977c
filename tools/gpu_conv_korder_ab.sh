#!/bin/bash
# Conv GEMMs' k-order A/B (MSH_CONV_KORDER=0 tap-major / 1 channel-block-major, read at LOAD): parity files, headline + conv2 / conv3
# times alternating on one box, FETCH_SIZE of both kernels under each order (HISTORY.md, round 6).
set -u
export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full_batch.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2; do
for KO in 0 1; do
  MSH_CONV_KORDER=$KO timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
ks = {k["kernel"]: k for k in d["kernels"]}
print("KORDER=$KO rep $rep:", d["value"], "serial", d["serial_steps"]["value"], "ids", d["config"]["ids_match_serial_pass"],
      "conv2 ms", ks["conv2_gelu_gemm"]["ms_per_launch"], ks["conv2_gelu_gemm"]["frac"], "conv3 ms", ks["conv3_gelu_gemm"]["ms_per_launch"], ks["conv3_gelu_gemm"]["frac"])
PY
done
done
for KO in 0 1; do
  echo "== FETCH_SIZE, MSH_CONV_KORDER=$KO"
  MSH_CONV_KORDER=$KO bash tools/gpu_pmc_kernel.sh conv$KO "EpiGnBiasGeluBf16|EpiBiasGeluF32" FETCH_SIZE 2>&1 | tail -6
done
