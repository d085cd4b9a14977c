#!/bin/bash
# Round 5, GPU call 10: the single clip's encoder on the split-K GEMMs: parity files that encode few rows, then latency on / off
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5j}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dec_small.py tests/test_gpu_long_parity.py tests/test_gpu_panel.py tests/test_gpu_mlp.py -m gpu -q --durations=5 > gpurun_out/${TAG}_pytest.log 2>&1
tail -6 gpurun_out/${TAG}_pytest.log
{
for r in 1024 0; do echo "== MSH_ENC_SMALL_ROWS=$r"; MSH_ENC_SMALL_ROWS=$r timeout 300 python tools/latency_probe.py 2>&1 | grep "latency\|encoder"; done
} 2>&1 | tee gpurun_out/${TAG}_latency.txt
