#!/bin/bash
# Experiment: encoder attention workgroup shape (2 x 4 waves sharing a CU vs 1 x 7 waves), per-kernel time from bench's table
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --steps 12 --warmup 2"
for w in 0 1; do
  echo "== MSH_ENC_ATT_WIDE=$w"
  MSH_ENC_ATT_WIDE=$w timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'serial', d.get('serial_steps'))
for k in d['kernels']:
    if k['kernel'].startswith('enc') or 'attn' in k['kernel']: print('  ', k['kernel'], k.get('ms_per_launch'), k.get('frac'))
"
done
