#!/bin/bash
# Round 5, GPU call 9: looped cross-attention with the merges behind the loop (tests + chain at 8 / 16 / 63 clips), and where the
# single clip's encoder spends its 1.25 ms
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5i}
timeout 900 python -m pytest tests/test_gpu_dec_small.py -m gpu -q -s > gpurun_out/${TAG}_pytest.log 2>&1
tail -3 gpurun_out/${TAG}_pytest.log
{
timeout 300 python tools/chain_probe.py 8 16 63 2>&1 | grep "^B="
timeout 300 python tools/latency_probe.py 2>&1 | tail -3
} 2>&1 | tee gpurun_out/${TAG}_chain.txt
