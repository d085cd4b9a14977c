#!/bin/bash
# The rolling batch call (default VAD options): its GPU tests, then the phase times of the C-API batch call with and without it.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-rolling}
export MSH_DEV_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_silero.py tests/test_gpu_capi.py tests/test_gpu_capi_threads.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
STEPS=${STEPS:-4} bash tools/gpu_rolling2.sh $TAG "${@:2}"
