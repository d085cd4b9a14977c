#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_capi.py -q -x -k "cross_attention_option" 2>&1 | tail -40 | cut -c1-240 | tee gpurun_out/r5h_capi_cross.txt
