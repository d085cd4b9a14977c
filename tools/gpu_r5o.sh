#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
timeout 300 python tools/stream_host_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rd5o_stream_host_probe.txt | tail -24
