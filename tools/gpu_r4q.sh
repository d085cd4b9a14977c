#!/bin/bash
# round 4: lanes-in-flight sweep with the absorbed cross-attention; ablations / timeline of that kernel
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4q}
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
{
for F in 4 5 6 8; do
  timeout 300 python bench.py --in-flight $F $FLAGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight $F', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'))"
done
} 2>&1 | tee gpurun_out/${TAG}_lanes.txt
{
timeout 120 python tools/xattn_microbench.py
MSH_XATTN_TIMELINE=1 timeout 120 python tools/xattn_microbench.py
for A in 1 2 3 4 6 7 10 11; do MSH_XATTN_ABL=$A timeout 120 python tools/xattn_microbench.py; done
for M in 128 512; do XA_M=$M timeout 120 python tools/xattn_microbench.py; done
} 2>&1 | tee gpurun_out/${TAG}_xattn.txt
