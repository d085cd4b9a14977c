#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capi.py -q -x 2>&1 | tail -1
timeout 300 python tools/chain_probe.py 1 16 48 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r5j_fused_q.txt
