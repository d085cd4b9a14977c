#!/bin/bash
# Round 4: absorbed cross-attention kernel after the 3-slot ring / buffer-load DMA / grouped LDS reads
set -u
TAG=${1:-r4b}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
for SL in 3 2; do
  MSH_XATTN_SLOTS=$SL timeout 600 python -m pytest tests/test_gpu_xattn.py -q -s -k "kernel" > gpurun_out/${TAG}_kernel_s$SL.log 2>&1; grep -E "absorbed cross|passed|failed|Error" gpurun_out/${TAG}_kernel_s$SL.log | tail -4
done
MSH_XATTN_TR=0 timeout 600 python -m pytest tests/test_gpu_xattn.py -q -s -k "kernel" > gpurun_out/${TAG}_kernel_tr0.log 2>&1; grep -E "absorbed cross|passed|failed|Error" gpurun_out/${TAG}_kernel_tr0.log | tail -4
timeout 1200 python -m pytest tests/test_gpu_xattn.py -q -s -k "not kernel" > gpurun_out/${TAG}_engine.log 2>&1; tail -4 gpurun_out/${TAG}_engine.log
FLAGS="--steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
for SL in 3 2; do
  MSH_XATTN_SLOTS=$SL timeout 600 python bench.py $FLAGS > gpurun_out/${TAG}_bench_absorbed_s$SL.json 2> gpurun_out/${TAG}_bench_absorbed_s$SL.err; cut -c1-200 gpurun_out/${TAG}_bench_absorbed_s$SL.json
done
MSH_XATTN_MIN_BATCH=100000 timeout 600 python bench.py $FLAGS > gpurun_out/${TAG}_bench_classic.json 2> gpurun_out/${TAG}_bench_classic.err; cut -c1-200 gpurun_out/${TAG}_bench_classic.json
