#!/bin/bash
# run-to-run variance of the encoder kernel times (event scope) on ONE box: short bench x2, then with the sub-runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2 value', d['value'], ' '.join('%s=%.3f'%(k['kernel'].replace('enc_',''),k['ms_per_launch']) for k in d['kernels'] if k['kernel'] in ('enc_oproj_mlp_fused','enc_qkv_panel','enc_attention','conv2_gelu_gemm','cross_kv_gemm')))
"; }
S="--steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
timeout 300 python bench.py $S > /tmp/a.json 2>/dev/null; show /tmp/a.json short1
timeout 300 python bench.py $S > /tmp/b.json 2>/dev/null; show /tmp/b.json short2
timeout 600 python bench.py --no-cpu-baseline > /tmp/c.json 2>/dev/null; show /tmp/c.json full
timeout 300 python bench.py $S > /tmp/d.json 2>/dev/null; show /tmp/d.json short3
timeout 300 python tools/panel_microbench.py 0 2>&1 | tail -1
