#!/bin/bash
# round 4: single-clip path -- the fused-q cross-attention requests its first K rows before it forms the query and reduces by DPP
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capi.py tests/test_gpu_kv_fp8.py -q -x 2>&1 | tail -2
timeout 300 python tools/chain_probe.py 1 16 32 2>&1 | grep -v amdgpu.ids
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['latency_ms'])"
} 2>&1 | tee gpurun_out/r5g_single_clip.txt
