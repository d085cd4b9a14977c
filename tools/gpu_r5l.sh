#!/bin/bash
# Round 5, GPU call 12: tests touched by the development-library split and the 8-entry device lists; the two disputed probes
# redone (mfma_peak with shader clock + zero data, flag_handover with a resident grid); the slow-process hunt, bounded (6 processes)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-r5l}
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_panel.py tests/test_gpu_xattn.py tests/test_gpu_guard.py tests/test_gpu_capi.py tests/test_gpu_capi_streaming.py tests/test_gpu_dec_small.py -m gpu -q --durations=4 > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
(rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -12) > gpurun_out/${TAG}_smi.txt
timeout 120 tools/build/mfma_peak 2>&1 | tee gpurun_out/${TAG}_mfma_issue_ceiling.txt | cut -c1-250
timeout 120 tools/build/flag_handover 2>&1 | tee gpurun_out/${TAG}_flag_handover.txt | cut -c1-200
: > gpurun_out/${TAG}_slow_process.txt
for i in 1 2 3 4 5 6; do
  { echo "== process $i  $(rocm-smi --showclocks 2>/dev/null | grep -i 'sclk' | head -1 | tr -s ' ')  $(rocm-smi --showpower 2>/dev/null | grep -i 'power' | head -1 | tr -s ' ')";
    MSH_CHAIN_MASKS=0x3c,0xc3,0xff timeout 200 python tools/chain_masks.py 256 2>&1 | grep "round 1\|mask" | tail -4; } >> gpurun_out/${TAG}_slow_process.txt
done
grep "0xff\|== process" gpurun_out/${TAG}_slow_process.txt | cut -c1-220
