#!/bin/bash
# Evidence run of the round's last tree: the whole GPU suite, smoke(), the default bench line.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-final2}
MSH_DEV_KNOBS=1 timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench.json
