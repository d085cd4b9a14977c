#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
FLAGS="--in-flight 1 --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_serial -o t -- python $R/bench.py $FLAGS > /tmp/tr_serial.log 2>&1)
tail -1 /tmp/tr_serial.log | cut -c1-200
f=$(find /tmp/tr_serial -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/trace_gaps.py "$f" 2>&1 | tee gpurun_out/r4n_serial_gaps.txt
