#!/bin/bash
# Round 5, GPU call 14: the default bench line on the current tree (pre-final check of every sub-run) + the fused MLP's ablations
# with the weight-stream-only variant
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-rd5m}
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
timeout 300 python tools/mlp_microbench.py 2>&1 | tee gpurun_out/${TAG}_mlp_fused_ablations.txt | tail -14
