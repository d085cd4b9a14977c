#!/bin/bash
# Round 5, GPU call 2: the whole GPU suite on the tree with one cross-attention form per transcriber (no -x: every failure shows).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5b2}
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/${TAG}_pytest.log 2>&1
tail -30 gpurun_out/${TAG}_pytest.log
cp gpurun_out/parity_margins.json gpurun_out/${TAG}_parity_margins.json 2>/dev/null
