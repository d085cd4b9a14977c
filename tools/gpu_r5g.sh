#!/bin/bash
# Round 5, GPU call 7: latency path with the split cross-attention for every batch below 64 and the bit-identical fused
# self-attention: tests (clip alone == clip in a batch), chain costs at 1 / 8 / 16 / 32 / 48 / 63 clips split on and off
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5g}
timeout 900 python -m pytest tests/test_gpu_dec_small.py tests/test_gpu_capi.py -m gpu -q -s --durations=5 > gpurun_out/${TAG}_pytest.log 2>&1
tail -6 gpurun_out/${TAG}_pytest.log
{
for m in 4 0; do
  echo "== MSH_XSPLIT_M=$m"
  MSH_XLOOP=$([ $m = 0 ] && echo 0 || echo 1) MSH_XSPLIT_M=$m timeout 300 python tools/chain_probe.py 1 4 8 16 32 63 2>&1 | grep "^B="
done
echo "== latency, defaults"; timeout 300 python tools/latency_probe.py 2>&1 | tail -2
} 2>&1 | tee gpurun_out/${TAG}_chain.txt
