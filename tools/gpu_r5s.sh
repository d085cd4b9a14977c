#!/bin/bash
# round 4: conv1 / conv2 outputs with non-temporal stores (MSH_STEM_STORE_NT=1) -- overlapped bench, encoder kernel times
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
{
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or ragged" 2>&1 | tail -1
MSH_STEM_STORE_NT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or ragged" 2>&1 | tail -1
for V in 0 1 0 1; do
  MSH_STEM_STORE_NT=$V timeout 300 python bench.py $FLAGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stem store nt=$V', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), 'ids ok', d['config'].get('ids_match_serial_pass'))
print('   ', '  '.join('%s=%.4f' % (k['kernel'].replace('_gemm',''), k['ms_per_launch']) for k in d['kernels'] if k['kernel'].startswith(('conv','groupnorm'))))"
done
} 2>&1 | tee gpurun_out/r5s_stem_store_nt.txt
