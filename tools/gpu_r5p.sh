#!/bin/bash
# round 4: non-temporal policy on the streams a decode step reads once (self-attention K / V cache; the encoder rows of the
# absorbed cross-attention) -- does keeping them out of the memory-side cache make the layer's kernels cheaper IN SEQUENCE?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
export MSH_CHAIN_MASKS=0x03,0xff
{
for V in "0 0" "1 0" "0 256" "1 256" "0 0" "1 0"; do
  set -- $V
  echo "== MSH_SELF_NT=$1 MSH_XATTN_ABL=$2"
  MSH_SELF_NT=$1 MSH_XATTN_ABL=$2 timeout 300 python tools/chain_masks.py 2>&1 | grep -v amdgpu.ids | grep -v "round 1" | head -3 | cut -c1-200
done
unset MSH_CHAIN_MASKS
FLAGS="--steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
for V in "0 0" "1 0" "1 256" "0 0" "1 0"; do
  set -- $V
  MSH_SELF_NT=$1 MSH_XATTN_ABL=$2 timeout 300 python bench.py $FLAGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('self_nt=$1 xattn_abl=$2', d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), 'ids ok', d['config'].get('ids_match_serial_pass'))"
done
} 2>&1 | tee gpurun_out/r5p_nt_streams.txt
