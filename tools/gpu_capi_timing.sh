#!/bin/bash
# Phase times of the C-API batch call (MSH_HOST_TIMING=1) under vad_threshold = 0 and under the reference's default options.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-capi}
MSH_HOST_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-fp8 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_host_timing.txt
grep -i "batch call\|moonshine\]" gpurun_out/${TAG}_host_timing.txt | tail -30 | cut -c1-260
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['c_api_batch'])[:500])"
