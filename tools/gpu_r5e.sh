#!/bin/bash
# Round 5, GPU call 5: the 4-wave split cross-attention kernel + metadata-free merge: tests, then batch-1 latency and chain costs
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5e}
timeout 900 python -m pytest tests/test_gpu_dec_small.py tests/test_gpu_parity.py -m gpu -q -s --durations=5 > gpurun_out/${TAG}_pytest.log 2>&1
tail -8 gpurun_out/${TAG}_pytest.log
for m in 8 0; do
  echo "== MSH_XSPLIT_M=$m"
  MSH_XSPLIT_M=$m timeout 300 python tools/latency_probe.py 2>&1 | tail -2
done 2>&1 | tee gpurun_out/${TAG}_latency.txt
