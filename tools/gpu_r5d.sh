#!/bin/bash
# Round 5, GPU call 4: split cross-attention after the P.V reduction fix -- its tests, the fused-argmax decode head, ablations of
# the split kernel (MSH_XSPLIT_ABL bits: 1 no Wq / MFMA, 2 no LayerNorm, 4 nothing behind the query)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5d}
timeout 900 python -m pytest tests/test_gpu_dec_small.py tests/test_gpu_parity.py -m gpu -q -s -x --durations=5 > gpurun_out/${TAG}_pytest.log 2>&1
tail -8 gpurun_out/${TAG}_pytest.log
for a in 0 1 2 4 7; do
  echo "== MSH_XSPLIT_M=8 MSH_XSPLIT_ABL=$a"
  MSH_XSPLIT_ABL=$a MSH_XSPLIT_M=8 timeout 300 python tools/latency_probe.py 2>&1 | tail -2
done 2>&1 | tee gpurun_out/${TAG}_latency.txt
