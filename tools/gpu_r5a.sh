#!/bin/bash
# Round 5, GPU call 1: network probe; the parity package (all 256 clips x 65 steps vs HF fp32 for both cross-attention forms,
# one-form-per-engine tests, graph cache test, capi form test); what a ragged sequence of batch shapes costs with the graph cache.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5a}
bash tools/gpu_netprobe.sh $TAG
timeout 900 python -m pytest tests/test_gpu_parity_full_batch.py tests/test_gpu_xattn.py tests/test_gpu_capi.py tests/test_gpu_parity.py -m gpu -q -x -s --durations=8 > gpurun_out/${TAG}_pytest.log 2>&1
tail -25 gpurun_out/${TAG}_pytest.log
cp gpurun_out/parity_margins.json gpurun_out/${TAG}_parity_margins.json 2>/dev/null
timeout 300 python tools/ragged_graph_probe.py 2>&1 | tee gpurun_out/${TAG}_ragged_graph_cache.txt
FLAGS="--steps 12 --warmup 2 --no-cpu-baseline --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
timeout 400 python bench.py $FLAGS > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
