#!/bin/bash
# Round 4, first run of the absorbed cross-attention (k_xattn.hip): the kernel alone with and without the transposing LDS
# read, the parity bodies with the form forced on, and a short bench line with the form on / off.
set -u
TAG=${1:-r4a}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_xattn.py -q -s -k "kernel" > gpurun_out/${TAG}_kernel_tr1.log 2>&1; tail -5 gpurun_out/${TAG}_kernel_tr1.log
MSH_XATTN_TR=0 timeout 600 python -m pytest tests/test_gpu_xattn.py -q -s -k "kernel" > gpurun_out/${TAG}_kernel_tr0.log 2>&1; tail -5 gpurun_out/${TAG}_kernel_tr0.log
timeout 1200 python -m pytest tests/test_gpu_xattn.py -q -s -k "not kernel" > gpurun_out/${TAG}_engine.log 2>&1; tail -8 gpurun_out/${TAG}_engine.log
FLAGS="--steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
timeout 600 python bench.py $FLAGS > gpurun_out/${TAG}_bench_absorbed.json 2> gpurun_out/${TAG}_bench_absorbed.err; cut -c1-300 gpurun_out/${TAG}_bench_absorbed.json
MSH_XATTN_MIN_BATCH=100000 timeout 600 python bench.py $FLAGS > gpurun_out/${TAG}_bench_classic.json 2> gpurun_out/${TAG}_bench_classic.err; cut -c1-300 gpurun_out/${TAG}_bench_classic.json
