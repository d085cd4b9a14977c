#!/bin/bash
# The rolling batch call (default VAD options): its GPU tests, then the phase times of the C-API batch call with and without it.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-rolling}
export MSH_DEV_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_silero.py tests/test_gpu_capi.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
bash tools/gpu_rolling2.sh $TAG X=1 MSH_BATCH_ROLLING=0 MSH_ROLLING_SHORT_FRAC=0.25
