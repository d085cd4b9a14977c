#!/bin/bash
# The small-batch decode path: its tests, then chain costs at the given batch sizes (default 8 16 32 63) and the batch-1 latency.
set -u
export MSH_DEV_KNOBS=1   # the library reads its developer switches only with this set
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-small}; shift || true
timeout 600 python -m pytest tests/test_gpu_dec_small.py -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -2 gpurun_out/${TAG}_pytest.log
timeout 300 python tools/chain_probe.py ${@:-8 16 32 63} 2>&1 | grep "^B=" | tee gpurun_out/${TAG}_chain.txt | cut -c1-200
