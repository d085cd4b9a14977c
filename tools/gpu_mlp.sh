#!/bin/bash
# The fused encoder MLP kernel: its parity tests (kernel alone vs numpy, encoder vs oracle) and the microbenchmark with ablations.
set -u
export MSH_DEV_KNOBS=1   # the library reads its developer switches only with this set
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-mlp}
timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_panel.py -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
timeout 300 python tools/mlp_microbench.py 2>&1 | tee gpurun_out/${TAG}_mlp_fused_ablations.txt
