#!/bin/bash
# round 4: encoder block kernels (fused MLP, QKV panel) with non-temporal output stores when lanes share the GPU
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
{
for V in 1 0 1 0; do
  MSH_ENC_STORE_NT=$V timeout 200 python bench.py $FLAGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('enc store nt=$V', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), 'ids ok', d['config'].get('ids_match_serial_pass'))
print('   ', '  '.join('%s=%.4f' % (k['kernel'], k['ms_per_launch']) for k in d['kernels'] if k['kernel'].startswith('enc_')))"
done
} 2>&1 | tee gpurun_out/r5v_enc_store_nt.txt
