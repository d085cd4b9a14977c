#!/bin/bash
# round 4: absorbed cross-attention after the merge rewrite (partials over the wave's own ring, one barrier, XCD-grouped
# clips) and with a second ring slot for waves 0-3 (MSH_XATTN_CFG=84, default) against one slot each (81)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4r}
{
for C in 84 81; do MSH_XATTN_CFG=$C timeout 300 python -m pytest tests/test_gpu_xattn.py -q -x -k "kernel" 2>&1 | tail -2; done
for C in 84 81 84 81; do MSH_XATTN_CFG=$C timeout 120 python tools/xattn_microbench.py; done
MSH_XATTN_XCD=0 timeout 120 python tools/xattn_microbench.py
MSH_XATTN_TIMELINE=1 timeout 120 python tools/xattn_microbench.py
for A in 2 4 6 10; do MSH_XATTN_ABL=$A timeout 120 python tools/xattn_microbench.py; done
for M in 64 128 192 512; do XA_M=$M timeout 120 python tools/xattn_microbench.py; done
for T in 100 250 830; do XA_T=$T timeout 120 python tools/xattn_microbench.py; done
} 2>&1 | tee gpurun_out/${TAG}_xattn.txt
FLAGS="--steps 16 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
timeout 300 python bench.py $FLAGS > gpurun_out/${TAG}_bench_quick.json 2>/dev/null
python -c "import sys,json; d=json.loads(open('gpurun_out/${TAG}_bench_quick.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), d['decode_step_us'])"
