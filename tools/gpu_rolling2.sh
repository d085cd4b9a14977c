#!/bin/bash
# Phase times of the rolling batch call under a few developer switches (same box).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-rolling}
export MSH_DEV_KNOBS=1
B="--steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-fp8"
i=0
for E in "${@:2}"; do
  i=$((i+1))
  env $E MSH_HOST_TIMING=1 timeout 600 python bench.py $B > gpurun_out/${TAG}_${i}_bench.json 2> gpurun_out/${TAG}_${i}_host_timing.txt
  echo "== $i: $E"
  grep -i "batch call: [0-9].*segment" gpurun_out/${TAG}_${i}_host_timing.txt | tail -2 | cut -c60-330
  grep "device VAD:" gpurun_out/${TAG}_${i}_host_timing.txt | tail -3 | cut -c60-300
  python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_${i}_bench.json').read().strip().splitlines()[-1]); c=d['c_api_batch']; print(c['value'], c['ms_per_call'], '| default_vad', c['default_vad']['value'], c['default_vad']['ms_per_call'])"
done
