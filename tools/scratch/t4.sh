export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -12
