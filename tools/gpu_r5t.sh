#!/bin/bash
# round 4: is the 4 -> 5 lanes cliff the pool of hardware queues (GPU_MAX_HW_QUEUES = 8 in bench.py)?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
FLAGS="--steps 20 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
{
for V in "8 4" "16 4" "16 5" "16 6" "24 6" "24 8" "16 5"; do
  set -- $V
  GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --in-flight $2 $FLAGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hw queues $1 lanes $2:', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), 'ids ok', d['config'].get('ids_match_serial_pass'))"
done
} 2>&1 | tee gpurun_out/r5t_hw_queues.txt
