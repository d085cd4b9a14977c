#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
tools/build/mfma_peak 2>&1 | tee gpurun_out/r5o_mfma_peak.txt
tools/build/mfma_peak 2>&1 | tail -11 >> gpurun_out/r5o_mfma_peak.txt
