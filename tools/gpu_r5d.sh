#!/bin/bash
# round 4: encoder attention without a branch per P.V MFMA (default) and with two query tiles per wave (MSH_ENC_ATT_EQT=2);
# decode self-attention with DPP reductions
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_parity.py -q -x 2>&1 | tail -2
MSH_ENC_ATT_EQT=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 --steps 12 --warmup 2"
for e in 4 2 4 2; do
  echo "== MSH_ENC_ATT_EQT=$e"
  MSH_ENC_ATT_EQT=$e timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'))
for k in d['kernels']:
    if k['kernel'] in ('enc_attention','dec_self_attention','enc_qkv_panel','enc_oproj_mlp_fused'): print('  ', k['kernel'], k.get('ms_per_launch'), k.get('frac'))
"
done
} 2>&1 | tee gpurun_out/r5d_attention.txt
