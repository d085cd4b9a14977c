#!/bin/bash
# Experiment: decode groups (the batch's rows split over parallel decode chains on their own streams) after the round-2
# decode kernels; plus the GPU suite on the current build.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --steps 12 --warmup 2"
for g in 1 2 3 4; do
  echo "== MSH_DEC_GROUPS=$g in-flight 4"
  MSH_DEC_GROUPS=$g timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'serial', d.get('serial_steps'))"
done
for g in 2; do for f in 2 3; do
  echo "== MSH_DEC_GROUPS=$g in-flight $f"
  MSH_DEC_GROUPS=$g timeout 300 $B --in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'serial', d.get('serial_steps'))"
done; done
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/exp_pytest.log 2>&1; tail -3 gpurun_out/exp_pytest.log
