#!/bin/bash
# rocprofv3 kernel stats of the C-API batch call (vad_threshold 0 and the reference's default options) as bench.py runs it.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-capi_prof}
export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-fp8"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- $CMD > /tmp/prof_${TAG}.log 2>&1)
f=$(find /tmp/prof_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_rocprofv3_kernel_stats_c_api_batch.csv && head -30 "$f" | cut -c1-150
tail -c 700 /tmp/prof_${TAG}.log
