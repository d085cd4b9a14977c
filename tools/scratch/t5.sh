export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_capi.py tests/test_gpu_capi_threads.py tests/test_gpu_capi_streaming.py tests/test_gpu_dist.py -m gpu -q 2>&1 | grep -v "^\[moonshine" | tail -30
