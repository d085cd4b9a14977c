#!/bin/bash
# kernel trace of the overlapped bench (4 lanes) and what is resident when (tools/trace_concurrency.py)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4k}
export TMPDIR=/tmp
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
for MODE in absorbed classic; do
  EXTRA=""; [ $MODE = classic ] && EXTRA="MSH_XATTN_MIN_BATCH=100000"
  (cd /tmp && env $EXTRA MSH_XATTN_G2_TN=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$MODE -o t -- python $R/bench.py $FLAGS > /tmp/tr_$MODE.log 2>&1)
  f=$(find /tmp/tr_$MODE -name "*kernel_trace.csv" | head -1)
  echo "== $MODE"; tail -1 /tmp/tr_$MODE.log | cut -c1-160
  [ -n "$f" ] && python tools/trace_concurrency.py "$f"
done 2>&1 | tee gpurun_out/${TAG}_concurrency.txt
