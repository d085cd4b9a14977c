#!/bin/bash
# placement sensitivity of the encoder panel kernels: shift the engine's allocations by a dummy block and time the kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="--steps 6 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
for kb in ${SKEWS:-0 64 1024 2048 3072 4096 5408 6144 8192 16384}; do
  env ${SKEWVAR:-MSH_ALLOC_SKEW_KB}=$kb timeout 300 python bench.py $S 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skew %6d KB value'%$kb, d['value'], ' '.join('%s=%.3f'%(k['kernel'].replace('enc_',''),k['ms_per_launch']) for k in d['kernels'] if k['kernel'] in ('enc_oproj_mlp_fused','enc_qkv_panel','enc_attention','conv2_gelu_gemm','cross_kv_gemm')))
"
done
