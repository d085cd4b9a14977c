#!/bin/bash
# Round 5, GPU call 3: the single-clip latency path (split cross-attention) -- its tests, the capi / long-parity files that
# failed under the old auto rule, the batch-1 chain profile and the latency bench, split on and off.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5c}
timeout 900 python -m pytest tests/test_gpu_dec_small.py tests/test_gpu_capi.py tests/test_gpu_long_parity.py::test_two_cities_wav_through_the_c_api "tests/test_gpu_streaming.py::test_cross_attention_runs_kernel_is_bit_identical_to_the_per_row_kernel" tests/test_gpu_parity.py -m gpu -q -s --durations=5 > gpurun_out/${TAG}_pytest.log 2>&1
tail -15 gpurun_out/${TAG}_pytest.log
for m in 8 0; do
  echo "== MSH_XSPLIT_M=$m"
  MSH_XSPLIT_M=$m timeout 300 python tools/latency_probe.py 2>&1 | tail -3
done 2>&1 | tee gpurun_out/${TAG}_latency.txt
