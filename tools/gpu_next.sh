#!/bin/bash
# First GPU call of the next round: the cache-policy probes that round 4 compiled but could not measure any more
# (DESIGN.md 3c / 9c), on the streaming workload (config 5) and, for reference, the offline headline.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-next}
{
for V in "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $V
  MSH_STREAM_SELF_NT=$1 MSH_STREAM_XRUNS_NT=$2 timeout 120 python bench.py --workload streaming --steps 2 --warmup 1 --no-stream-profile 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream self nt=$1 runs nt=$2:', d['value'], d.get('ms_per_step'))"
done
MSH_STREAM_SELF_NT=1 MSH_STREAM_XRUNS_NT=1 timeout 300 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_capi_streaming.py -q -x 2>&1 | tail -1
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
timeout 300 python bench.py $FLAGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('offline', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'))"
} 2>&1 | tee gpurun_out/${TAG}_cache_policy_probes.txt
