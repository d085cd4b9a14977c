#!/bin/bash
# The whole GPU suite + smoke on the current tree (no profiles): tools/gpu_final.sh is the evidence run.
set -u
export MSH_DEV_KNOBS=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-suite}
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/${TAG}_pytest.log 2>&1; tail -9 gpurun_out/${TAG}_pytest.log
timeout 300 env -u MSH_DEV_KNOBS python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
