#!/bin/bash
# round 4: re-run of r5c / r5d / r5g on a library that really contains the changes (the earlier runs used a stale .so: a
# compile error in k_xattn.hip had stopped the link)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
{
strings moonshine_amd/lib/libmoonshine.so | grep -c "cross_attention must be"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capi.py tests/test_gpu_kv_fp8.py -q -x 2>&1 | tail -3
MSH_ENC_ATT_EQT=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -1
MSH_XATTN_ABL=256 timeout 300 python -m pytest tests/test_gpu_xattn.py -q -x -k "kernel" 2>&1 | tail -1
for A in 0 256 0 256; do MSH_XATTN_ABL=$A timeout 120 python tools/xattn_microbench.py; done
timeout 300 python tools/chain_probe.py 1 16 2>&1 | grep -v amdgpu.ids
B="python bench.py --no-cpu-baseline --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 --steps 12 --warmup 2"
for e in 4 2 4; do
  echo "== MSH_ENC_ATT_EQT=$e"
  MSH_ENC_ATT_EQT=$e timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), d['latency_ms'])
for k in d['kernels']:
    if k['kernel'] in ('enc_attention','dec_self_attention','enc_qkv_panel','enc_oproj_mlp_fused','dec_crossq_gemm'): print('  ', k['kernel'], k.get('ms_per_launch'), k.get('frac'))
"
done
} 2>&1 | tee gpurun_out/r5i_rerun.txt
