#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
timeout 300 python tools/chain_probe.py 1 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5f_chain_small_batch.txt
tools/build/launch_floor 2>&1 | tee gpurun_out/r5f_launch_floor.txt
