#!/bin/bash
# Round 5, GPU call 11: the whole GPU suite on the latency-path tree; decode steps per graph replay at one clip
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5k}
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/${TAG}_pytest.log 2>&1
tail -12 gpurun_out/${TAG}_pytest.log
cp gpurun_out/parity_margins.json gpurun_out/${TAG}_parity_margins.json 2>/dev/null
{
for n in 4 8 16 32; do echo "== MSH_DEC_GRAPH_STEPS=$n"; MSH_DEC_GRAPH_STEPS=$n timeout 300 python tools/latency_probe.py 2>&1 | grep "latency"; done
} 2>&1 | tee gpurun_out/${TAG}_graph_steps.txt
