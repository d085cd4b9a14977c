#!/bin/bash
# round 4: do the encoder panel kernels (and everything else) care where the workspaces start relative to each other?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 --steps 12 --warmup 2"
{
for k in 0 1 4 17 64 0 4; do
  echo "== MSH_BUF_SKEW_KB=$k"
  MSH_BUF_SKEW_KB=$k timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), 'ids ok', d['config'].get('ids_match_serial_pass'))
print('  ', '  '.join('%s=%.4f' % (k['kernel'].replace('enc_','').replace('_gemm',''), k['ms_per_launch']) for k in d['kernels'] if k['kernel'].startswith(('enc_','conv','cross_kv'))))
"
done
} 2>&1 | tee gpurun_out/r5m_buffer_skew.txt
