#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_streaming.py -q -x 2>&1 | tail -1 | tee gpurun_out/r5y_stream_tests.txt
