#!/bin/bash
# round 4: two-stage query kernel of the absorbed cross-attention (k_crossq.hip) -- kernel test, engine parity with it,
# bench with it and with the merged-weight GEMM it replaces (MSH_XATTN_QT=1)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4u}
{
timeout 600 python -m pytest tests/test_gpu_xattn.py -q -x -s 2>&1 | tail -12
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), d['decode_step_us'])"; }
timeout 300 python bench.py $FLAGS 2>/dev/null | one "two-stage"
MSH_XATTN_QT=1 timeout 300 python bench.py $FLAGS 2>/dev/null | one "merged"
timeout 300 python bench.py $FLAGS 2>/dev/null | one "two-stage"
} 2>&1 | tee gpurun_out/${TAG}_crossq2.txt
