#!/bin/bash
# Round 5, GPU call 6: latency path = split cross-attention (vectorised merge) + self-attention fused into the o-proj launch +
# fused-argmax decode head: tests, then batch-1 latency / chain costs with each piece switched off in turn
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5f}
timeout 900 python -m pytest tests/test_gpu_dec_small.py tests/test_gpu_parity.py tests/test_gpu_capi.py -m gpu -q -s --durations=5 > gpurun_out/${TAG}_pytest.log 2>&1
tail -8 gpurun_out/${TAG}_pytest.log
grep "split vs" gpurun_out/${TAG}_pytest.log
run() { echo "== $*"; env "$@" timeout 300 python tools/latency_probe.py 2>&1 | tail -2; }
{
run MSH_XSPLIT_M=8
run MSH_XSPLIT_M=8 MSH_SELF_FUSED_M=0
run MSH_XSPLIT_M=8 MSH_NO_FUSED_ARGMAX=1
run MSH_XSPLIT_M=0 MSH_SELF_FUSED_M=0 MSH_NO_FUSED_ARGMAX=1
} 2>&1 | tee gpurun_out/${TAG}_latency.txt
