#!/bin/bash
# round-4 re-entry: GPU suite + default bench line + serial rocprofv3 kernel stats on the current tree
set -u
TAG=${1:-r4p}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/${TAG}_pytest.log 2>&1; tail -15 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
CMD="python $R/bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- $CMD > /tmp/prof_${TAG}.log 2>&1)
f=$(find /tmp/prof_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_rocprofv3_kernel_stats_b256_serial.csv && head -12 "$f" | cut -c1-200
