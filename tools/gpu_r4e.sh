#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
{
timeout 300 python -m pytest tests/test_gpu_xattn.py -q -k "kernel" 2>&1 | tail -1
for C in 81 43 42; do MSH_XATTN_CFG=$C timeout 120 python tools/xattn_microbench.py; done
MSH_XATTN_TIMELINE=1 timeout 120 python tools/xattn_microbench.py
for M in 64 128 512; do XA_M=$M timeout 120 python tools/xattn_microbench.py; done
timeout 900 python -m pytest tests/test_gpu_xattn.py -q -s -k "not kernel" 2>&1 | tail -6
FLAGS="--steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
timeout 600 python bench.py $FLAGS > gpurun_out/r4h_bench_absorbed.json 2> gpurun_out/r4h_bench_absorbed.err; cut -c1-200 gpurun_out/r4h_bench_absorbed.json
} 2>&1 | tee gpurun_out/r4h_xattn_qf.txt
