#!/bin/bash
# round 4: non-temporal streams as the default policy (self-attention cache always, the absorbed cross-attention's encoder rows
# when the engine is one of several lanes) -- parity, lanes, and the LM head's weight as a further candidate
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_xattn.py tests/test_gpu_parity.py tests/test_gpu_capi.py tests/test_gpu_capi_threads.py -q -x 2>&1 | tail -2
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), 'ids ok', d['config'].get('ids_match_serial_pass'))"; }
timeout 300 python bench.py $FLAGS 2>/dev/null | one "default (self nt, xattn nt in lanes)"
MSH_LMHEAD_NT=1 timeout 300 python bench.py $FLAGS 2>/dev/null | one "+ LM head nt"
MSH_SELF_NT=0 MSH_XATTN_NT=0 timeout 300 python bench.py $FLAGS 2>/dev/null | one "no nt at all"
MSH_XATTN_NT=1 timeout 300 python bench.py $FLAGS 2>/dev/null | one "xattn nt everywhere"
timeout 300 python bench.py $FLAGS 2>/dev/null | one "default"
MSH_LMHEAD_NT=1 timeout 300 python bench.py $FLAGS 2>/dev/null | one "+ LM head nt"
for F in 5 6; do timeout 300 python bench.py --in-flight $F $FLAGS 2>/dev/null | one "default, in-flight $F"; done
} 2>&1 | tee gpurun_out/r5q_nt_default.txt
