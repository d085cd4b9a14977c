#!/bin/bash
# The guard-allocator run of tools/gpu_guard.sh for a few test files only: gpu_guard_files.sh TAG tests/test_gpu_x.py ...
set -u
export MSH_DEV_KNOBS=1 TMPDIR=/tmp MSH_GUARD_ALLOC=1
TAG=${1:-guardf}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for f in "${@:2}"; do
  n=$(basename "$f" .py)
  timeout 900 python -m pytest "$f" -m gpu -v -p no:cacheprovider > gpurun_out/${TAG}_$n.log 2>&1
  echo "guard $n rc=$?: $(tail -1 gpurun_out/${TAG}_$n.log | cut -c1-200)"
  grep "FAILED\|Memory access fault" gpurun_out/${TAG}_$n.log | cut -c1-250 | head -5
done
