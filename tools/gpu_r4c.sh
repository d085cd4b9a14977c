#!/bin/bash
# absorbed cross-attention kernel: workgroup shapes (MSH_XATTN_CFG = <waves><slots>) and ablations of the default one
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4c}
{
for C in 81 43 42 40; do
  MSH_XATTN_CFG=$C timeout 300 python -m pytest tests/test_gpu_xattn.py -q -k "kernel" 2>&1 | tail -1
  for M in 64 128 256 512; do MSH_XATTN_CFG=$C XA_M=$M timeout 120 python tools/xattn_microbench.py; done
done
for A in 1 2 4 6; do MSH_XATTN_ABL=$A timeout 120 python tools/xattn_microbench.py; done
for T in 100 200 830 1660; do XA_T=$T timeout 120 python tools/xattn_microbench.py; done
} 2>&1 | tee gpurun_out/${TAG}_xattn_shapes.txt
