#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
{
for A in 0 256 0 256 6 262; do MSH_XATTN_ABL=$A timeout 120 python tools/xattn_microbench.py; done
MSH_XATTN_ABL=256 timeout 300 python -m pytest tests/test_gpu_xattn.py -q -x -k "kernel" 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r5c_xattn_nt.txt
